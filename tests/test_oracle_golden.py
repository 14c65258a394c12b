"""Pins the CPU oracle (oracle/*.py) against tests/golden/*.npz — outputs of the REFERENCE's own code
(gill.layers.TextFcLayer, gill.models.GILLModel / GILL) captured by oracle/gen_golden.py.  CPU only."""
import os

import numpy as np
import pytest
import torch

from gill_amd import synth
from oracle import clip_ref, mapper_ref, opt_ref, pipeline_ref, scheduler_ref

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(name):
  return np.load(os.path.join(GOLD, name), allow_pickle=False)


def _bf16_weights(sd):
  return {k: v.bfloat16().float() for k, v in sd.items()}


@pytest.mark.parametrize("tag,in_dim", [("d768_b2", 768), ("d4096_b1", 4096)])
def test_mapper_oracle_matches_reference(tag, in_dim):
  g = _load(f"mapper_{tag}.npz")
  sd = _bf16_weights(synth.mapper_state_dict(synth.MapperConfig(in_dim=in_dim), seed=int(g["seed"])))
  y = mapper_ref.mapper_forward(sd, torch.from_numpy(g["x"]), torch.from_numpy(g["e"]))
  ref = torch.from_numpy(g["y"])
  assert y.shape == ref.shape == (g["x"].shape[0], 77, 768)
  assert (y - ref).abs().max().item() < 2e-4, (y - ref).abs().max().item()


@pytest.fixture(scope="module")
def opt125m_weights():
  cfg = synth.OptConfig(vocab_size=50274, hidden_size=768, num_layers=12, num_heads=12, ffn_dim=3072)
  return cfg, _bf16_weights(synth.opt_state_dict(cfg, seed=5)), _bf16_weights(synth.mapper_state_dict(synth.MapperConfig(in_dim=768), seed=7))


def test_img_hidden_states_oracle_matches_reference(opt125m_weights):
  """GILLModel.forward(mode='generation') on a ragged right-padded batch without attention mask."""
  cfg, osd, msd = opt125m_weights
  g = _load("gillmodel_forward_opt125m.npz")
  labels, cap = torch.from_numpy(g["labels"]), torch.from_numpy(g["caption_len"])
  raw, emb = pipeline_ref.img_hidden_and_embeds(osd, cfg.num_layers, cfg.num_heads, labels, cap - 1)
  ref_h = torch.from_numpy(g["llm_hidden"])
  assert (raw - ref_h).abs().max().item() < 5e-4, (raw - ref_h).abs().max().item()
  y = mapper_ref.mapper_forward(msd, raw, emb)
  ref_y = torch.from_numpy(g["last_embedding"])
  assert ((y - ref_y) ** 2).mean().item() < 1e-7
  assert (y - ref_y).abs().max().item() < 2e-3


def test_generate_loop_equals_single_pass(opt125m_weights):
  cfg, osd, _ = opt125m_weights
  g = _load("gillmodel_generate_opt125m.npz")
  assert g["gen_ids"].tolist() == [synth.IMG_TOKEN_IDS * 2]          # forced [IMG0..7], twice (num_words=2)
  assert float(g["loop_vs_single_maxabs"]) < 1e-4                    # the 2-step loop == one pass over prompt ++ [IMG]x8
  prompt = torch.from_numpy(g["prompt"])
  ids = torch.cat([prompt, torch.tensor([synth.IMG_TOKEN_IDS])], dim=1)
  hid = opt_ref.opt_hidden_states(osd, cfg.num_layers, cfg.num_heads, opt_ref.opt_embed(osd, ids))
  ref = torch.from_numpy(g["hidden_img"])
  assert (hid[:, -8:] - ref).abs().max().item() < 5e-4
  # step-0 logits (strided sample) of the tied lm_head
  hid0 = opt_ref.opt_hidden_states(osd, cfg.num_layers, cfg.num_heads, opt_ref.opt_embed(osd, prompt))
  logits = opt_ref.opt_logits(osd, hid0[:, -1])
  # the reference applied its in-place logit surgery before we sampled them: compare finite entries away from [IMG] ids
  ref_l = g["last_logits_step0"]
  mine = logits.numpy()[:, ::97]
  ok = np.isfinite(ref_l) & (np.arange(0, 50274, 97)[None, :] < 50265)
  assert np.abs(mine[ok] - ref_l[ok]).max() < 2e-3


def test_api_fixture_shape():
  g = _load("gill_api_opt125m.npz")
  assert str(g["caption"]) == " " + "".join(f"[IMG{i}]" for i in range(8))   # models.py:707, :759
  assert str(g["decision"]) == "['gen', [0, 1]]" and int(g["ret_len"]) == 0
  assert g["gen_emb"].shape == (1, 77, 768)


def test_scheduler_known_answers():
  s = scheduler_ref.PNDMSchedulerRef()
  ts = s.set_timesteps(50)
  assert len(ts) == 51 and ts[:4] == [981, 961, 961, 941] and ts[-1] == 1
  assert s.set_timesteps(10) == [901, 801, 801, 701, 601, 501, 401, 301, 201, 101, 1]
  ac = s.alphas_cumprod
  assert abs(float(ac[0]) - 0.99915) < 1e-6 and abs(float(ac[999]) - 0.0046602) < 2e-6
  # one PLMS trajectory on a constant eps: every multistep formula reduces to eps itself
  s.set_timesteps(10)
  x = torch.ones(1, 4, 2, 2)
  e = torch.full_like(x, 0.5)
  for t in s.timesteps:
    x = s.step(e, t, x)
  # closed form: x_T->x_0 under constant eps e: x = sqrt(a_p/a_t) x - (a_p - a_t) e / (a_t sqrt(1-a_p) + sqrt(a_t (1-a_t) a_p))
  y = torch.ones(1, 4, 2, 2)
  seq = [901, 801, 701, 601, 501, 401, 301, 201, 101, 1]
  for t in seq:
    p = t - 100
    a_t = ac[t]; a_p = ac[p] if p >= 0 else ac[0]
    y = (a_p / a_t) ** 0.5 * y - (a_p - a_t) * e / (a_t * (1 - a_p) ** 0.5 + (a_t * (1 - a_t) * a_p) ** 0.5)
  assert torch.allclose(x, y, atol=1e-5)


def test_visual_embs_oracle_matches_reference():
  """GILLModel.get_visual_embs(mode='captioning') of the reference (CLIPVisionModel pooler_output -> visual_embeddings)."""
  g = _load("gill_visual_tiny.npz")
  cfg = synth.ClipConfig.tiny()
  sd = _bf16_weights(synth.clip_state_dict(cfg, seed=int(g["clip_seed"])))
  proj = {}
  synth._linear(proj, "visual_embeddings", 4 * 768, cfg.hidden_size, int(g["clip_seed"]))
  proj = _bf16_weights(proj)
  ve = clip_ref.visual_embs(sd, proj["visual_embeddings.weight"], proj["visual_embeddings.bias"],
                            torch.from_numpy(g["pixel_values"]), cfg.patch_size, cfg.num_heads, 4)
  ref = torch.from_numpy(g["visual_embs"])
  assert ve.shape == ref.shape == (3, 4, 768)
  assert (ve - ref).abs().max().item() < 2e-5, (ve - ref).abs().max().item()


def test_clip_image_preprocessing_matches_transformers():
  """gill_amd.utils.ClipImageProcessor against the object the reference gets from AutoFeatureExtractor (gill/utils.py:113)."""
  transformers = pytest.importorskip("transformers")
  from PIL import Image
  from gill_amd.utils import ClipImageProcessor
  rng = np.random.default_rng(0)
  for (w, h, size) in [(640, 480, 224), (300, 500, 224), (100, 80, 224), (64, 48, 32), (31, 77, 32)]:
    img = Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8))
    ref = transformers.CLIPImageProcessor(size={"shortest_edge": size}, crop_size={"height": size, "width": size})(
      img, return_tensors="pt").pixel_values
    got = ClipImageProcessor(size, size)(img).pixel_values
    assert got.shape == ref.shape and (got - ref).abs().max().item() < 1e-5


def test_sd_driver_oracle_matches_reference_pipeline_call():
  """F8: the reference's own StableDiffusionPipeline.__call__ (gill/custom_sd.py:567-666) drove the oracle's UNet / PNDM scheduler
  / VAE; the oracle's restatement of that driver (pipeline_ref.denoise + vae_ref) must reproduce its latents and images
  exactly (same fp32 arithmetic, same call order), incl. per-sample negative embeddings, the warm-up repeat of the second
  timestep and the num_images_per_prompt expansion."""
  from oracle import vae_ref
  g = _load("sd_driver_tiny.npz")
  cfg, vcfg = synth.UNetConfig.tiny(16), synth.VAEConfig.tiny(16)
  usd = _bf16_weights(synth.unet_state_dict(cfg, seed=int(g["unet_seed"])))
  vsd = _bf16_weights(synth.vae_decoder_state_dict(vcfg, seed=int(g["vae_seed"])))
  cond, neg, lat0 = (torch.from_numpy(g[k]) for k in ("cond", "neg", "lat0"))
  lat = pipeline_ref.denoise(usd, cond, neg, lat0, int(g["steps"]), float(g["guidance"]), cfg.block_out_channels, cfg.num_heads,
                             cfg.norm_num_groups)
  ref = torch.from_numpy(g["latents"])
  assert (lat - ref).abs().max().item() <= 1e-4 * ref.abs().max().item(), (lat - ref).abs().max().item()
  img = vae_ref.vae_decode(vsd, lat, vcfg.block_out_channels, vcfg.norm_num_groups, vcfg.scaling_factor)
  img = (img / 2 + 0.5).clamp(0, 1).permute(0, 2, 3, 1)
  assert (img - torch.from_numpy(g["images"]).float()).abs().max().item() < 2e-3      # fixture stored as fp16
  # num_images_per_prompt = 2 of ONE prompt: both images share the prompt, each has its own latent; 3 steps = timesteps 667, 334, 334, 1
  assert g["unet_call_timesteps"].tolist() == scheduler_ref.PNDMSchedulerRef().set_timesteps(3)
  lat2 = pipeline_ref.denoise(usd, cond[:1].repeat(2, 1, 1), neg[:1].repeat(2, 1, 1), lat0, 3, float(g["guidance"]),
                              cfg.block_out_channels, cfg.num_heads, cfg.norm_num_groups)
  ref2 = torch.from_numpy(g["latents_n2"])
  assert (lat2 - ref2).abs().max().item() <= 1e-4 * ref2.abs().max().item()
