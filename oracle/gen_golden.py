"""Generates tests/golden/*.npz by RUNNING THE REFERENCE (kohjingyu/gill at /root/reference) on CPU.

Run once in the build container:   python oracle/gen_golden.py
Nothing of the reference travels: only seeded inputs and the reference's numeric outputs are stored (weights are
regenerated in the tests from gill_amd.synth's counter-based generator with the seeds recorded here).

What is executed from the reference, unmodified:
  F1  gill.layers.TextFcLayer(mode='gill_mapper').forward                       (gill/layers.py:28-53)
  F2  gill.models.GILLModel.forward(mode='generation')                          (gill/models.py:164-441)
  F3  gill.models.GILLModel.generate(..., gen_scale_factor=1e5)                 (gill/models.py:443-532)
  F4  gill.models.GILL(load_sd=False).generate_for_images_and_texts             (gill/models.py:582-762)
  F5  gill.models.GILLModel.get_visual_embs(mode='captioning')                  (gill/models.py:129-146)
  F6  generate_for_images_and_texts([PIL image, text])                          (gill/models.py:606-613)
  F7  the retrieval branch of the same method (emb_matrix / path_array given)  (gill/models.py:671-696)
  F9  GILLModel.get_visual_embs(mode='retrieval') + the CLIP rerank of generated images   (gill/models.py:141-146, 724-751)
  F8  gill.custom_sd.StableDiffusionPipeline.__call__ — the CFG / scheduler / decode DRIVER     (gill/custom_sd.py:567-666,
      with _encode_prompt :224-373, prepare_latents :458-473, decode_latents :385-392) — run with the ORACLE's UNet, PNDM
      scheduler and VAE decoder injected as self.unet / self.scheduler / self.vae.  This pins the order of operations of the
      loop (negative|positive concat, scale_model_input, guidance combine, scheduler.step call pattern, 1/0.18215, /2+0.5 clamp,
      NHWC float32) to the reference's own lines.  It does NOT pin the UNet / scheduler / VAE arithmetic, which stays a
      restatement of diffusers==0.17.1 (absent): see oracle/__init__.py.
Harness shim: `diffusers` / `torchvision` are absent here, so empty stand-in modules are placed in
sys.modules BEFORE importing gill.models (its stage-3 code is never called: load_sd=False); random-init OPT / CLIP
models are saved to local dirs whose paths contain 'facebook/opt' and 'clip' (string checks at models.py:56,78); a
stand-in tokenizer object (gill_amd.synth.HashTokenizer) replaces the GPT2 tokenizer, whose vocab files are not
available offline.  transformers here is 5.x (reference pins 4.30.2): hidden_states[-1] is post-final-LN in both.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
"""
import os
import shutil
import sys
import tempfile
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")

from gill_amd import synth  # noqa: E402


def _import_reference():
  import transformers  # noqa: F401  (must be imported before the stubs)
  for name in ("diffusers", "torchvision", "torchvision.transforms", "torchvision.transforms.functional"):
    if name not in sys.modules:
      sys.modules[name] = types.ModuleType(name)
  sys.modules["diffusers"].StableDiffusionPipeline = object
  sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
  sys.modules["torchvision.transforms"].functional = sys.modules["torchvision.transforms.functional"]
  sys.path.insert(0, REF)
  import gill.layers as ref_layers
  import gill.models as ref_models
  import gill.utils as ref_utils
  from transformers import CLIPImageProcessor
  ref_utils.get_feature_extractor_for_model = lambda name, **kw: CLIPImageProcessor()
  return ref_layers, ref_models


def golden_mapper(ref_layers):
  """F1: reference TextFcLayer on synth weights (weights bf16-rounded, as the GPU engine stores them)."""
  for in_dim, B, tag in ((768, 2, "d768_b2"), (4096, 1, "d4096_b1")):
    cfg = synth.MapperConfig(in_dim=in_dim)
    sd = {k: v.bfloat16().float() for k, v in synth.mapper_state_dict(cfg, seed=11).items()}
    layer = ref_layers.TextFcLayer(in_dim, 768, num_input_tokens=8, num_output_tokens=77, mode="gill_mapper")
    layer.load_state_dict(sd, strict=True)
    layer.eval()
    x = synth.normal(f"mapper_x_{tag}", (B, 8, in_dim), 11).bfloat16().float()
    e = synth.normal(f"mapper_e_{tag}", (1, 8, in_dim), 11, 0.5).bfloat16().float()
    with torch.no_grad():
      y = layer(x, e)
      y_b = layer(x, e.repeat(B, 1, 1))
    assert torch.equal(y, y_b)
    np.savez_compressed(os.path.join(OUT, f"mapper_{tag}.npz"), x=x.numpy(), e=e.numpy(), y=y.numpy(),
                        seed=np.int64(11), in_dim=np.int64(in_dim))
    print("F1", tag, tuple(y.shape), float(y.abs().mean()))


def golden_gillmodel(ref_models, tmp):
  """F2-F4 on an opt-125m-shaped random model (12 layers, D=768, 12 heads) with the resized 50274 vocabulary."""
  from transformers import CLIPVisionConfig, CLIPVisionModel, OPTConfig, OPTForCausalLM
  ocfg = synth.OptConfig.opt_125m()
  ocfg.vocab_size = 50272     # HF opt-125m vocab before resize_token_embeddings(50274) (models.py:73)
  opt_dir = os.path.join(tmp, "facebook/opt-125m-shape")
  clip_dir = os.path.join(tmp, "openai/clip-tiny")
  hf_cfg = OPTConfig(vocab_size=ocfg.vocab_size, hidden_size=ocfg.hidden_size, num_hidden_layers=ocfg.num_layers,
                     ffn_dim=ocfg.ffn_dim, num_attention_heads=ocfg.num_heads, max_position_embeddings=2048,
                     word_embed_proj_dim=ocfg.hidden_size, do_layer_norm_before=True, dropout=0.0)
  hf = OPTForCausalLM(hf_cfg)
  sd_full = synth.opt_state_dict(synth.OptConfig(vocab_size=50274, hidden_size=768, num_layers=12, num_heads=12,
                                                 ffn_dim=3072), seed=5)
  sd_full = {k: v.bfloat16().float() for k, v in sd_full.items()}
  sd_hf = dict(sd_full)
  sd_hf["model.decoder.embed_tokens.weight"] = sd_full["model.decoder.embed_tokens.weight"][:50272].clone()
  sd_hf["lm_head.weight"] = sd_hf["model.decoder.embed_tokens.weight"]
  hf.load_state_dict(sd_hf, strict=True)
  hf.save_pretrained(opt_dir)
  CLIPVisionModel(CLIPVisionConfig(hidden_size=64, intermediate_size=128, num_hidden_layers=1, num_attention_heads=2,
                                   image_size=32, patch_size=16)).save_pretrained(clip_dir)

  tok = synth.HashTokenizer()
  args = types.SimpleNamespace(freeze_lm=True, freeze_vm=True, opt_version=opt_dir, visual_encoder=clip_dir,
                               n_visual_tokens=4, ret_emb_dim=256, gen_emb_dim=768, text_emb_layers=[-1],
                               text_fc_mode="gill_mapper", ret_text_fc_mode="linear", num_tokens=8, num_clip_tokens=77,
                               retrieval_token_idx=synth.IMG_TOKEN_IDS, gen_token_idx=synth.IMG_TOKEN_IDS)
  gill = ref_models.GILL(tok, args, load_sd=False, num_gen_images=1)
  gm = gill.model
  # the two resized rows and the [IMG] rows: take them from the synth table (as load_gill does for [IMG] rows)
  with torch.no_grad():
    gm.input_embeddings.weight.copy_(sd_full["model.decoder.embed_tokens.weight"])
    mcfg = synth.MapperConfig(in_dim=768)
    msd = {k: v.bfloat16().float() for k, v in synth.mapper_state_dict(mcfg, seed=7).items()}
    gm.gen_text_hidden_fcs[0].load_state_dict(msd, strict=True)
  gill.eval()

  # ---- F2: batched forward, ragged right-padded batch (no attention mask: models.py:363-365)
  B, Tp = 3, 14
  ids = synth.synthetic_prompt_ids(B, Tp, seed=3)              # (B, Tp+8), last 8 = [IMG0..7]
  lens = [Tp + 8, Tp + 8 - 3, Tp + 8 - 6]
  T = Tp + 8
  labels = torch.full((B, T), tok.pad_token_id, dtype=torch.int64)
  for b in range(B):
    n_words = lens[b] - 8
    labels[b, :n_words] = ids[b, :n_words]
    labels[b, n_words:lens[b]] = torch.tensor(synth.IMG_TOKEN_IDS)
  caption_len = torch.tensor(lens, dtype=torch.int64)
  with torch.no_grad():
    out = gm(torch.zeros(B, 3, 32, 32), labels.clone(), caption_len, mode="generation")
  last_embedding, llm_hidden = out[2], out[7][0]
  np.savez_compressed(os.path.join(OUT, "gillmodel_forward_opt125m.npz"), labels=labels.numpy(), caption_len=caption_len.numpy(),
                      llm_hidden=llm_hidden.numpy(), last_embedding=last_embedding.numpy(), full_labels=out[1].numpy(),
                      opt_seed=np.int64(5), mapper_seed=np.int64(7))
  print("F2", tuple(llm_hidden.shape), tuple(last_embedding.shape), float(llm_hidden.abs().mean()), float(last_embedding.abs().mean()))

  # ---- F3: the generate() loop (2 steps, forced [IMG]) vs the single pass
  prompt = labels[0:1, :Tp]
  with torch.no_grad():
    emb = gm.input_embeddings(prompt)
    g_ids, g_embs, g_logits = gm.generate(emb, 2, gen_scale_factor=1e5)
  hid_loop = g_embs[-1][:, Tp:Tp + 8]
  delta = float((hid_loop - llm_hidden[0:1]).abs().max())
  np.savez_compressed(os.path.join(OUT, "gillmodel_generate_opt125m.npz"), prompt=prompt.numpy(), gen_ids=g_ids.numpy(),
                      hidden_img=hid_loop.numpy(), last_logits_step0=g_logits[0].numpy()[:, ::97].copy(),
                      loop_vs_single_maxabs=np.float64(delta))
  print("F3 ids", g_ids.tolist(), "loop-vs-single max|d| =", delta)

  # ---- F4: public API shape/values
  text = "a small red fox jumps over the lazy dog"
  with torch.no_grad():
    ret = gill.generate_for_images_and_texts([text], num_words=2, gen_scale_factor=1e5)
  assert isinstance(ret[0], str) and isinstance(ret[1], dict)
  gen = ret[1]["gen"][0]
  np.savez_compressed(os.path.join(OUT, "gill_api_opt125m.npz"), text=np.array(text), caption=np.array(ret[0]),
                      decision=np.array(str(ret[1]["decision"])), ret_len=np.int64(len(ret[1]["ret"])), gen_emb=gen.numpy())
  print("F4", repr(ret[0]), ret[1]["decision"], tuple(gen.shape))


def golden_visual(ref_models, tmp):
  """F5-F6: image prompts.  A second reference GILL whose CLIPVisionModel has gill_amd.synth.ClipConfig.tiny() shapes and
  synth weights: F5 = GILLModel.get_visual_embs(mode='captioning') (gill/models.py:129-146), F6 = the public
  generate_for_images_and_texts([PIL image, text]) (gill/models.py:606-613 + the 'gen' branch)."""
  import gill.utils as ref_utils
  from PIL import Image
  from transformers import CLIPImageProcessor, CLIPVisionConfig, CLIPVisionModel, OPTConfig, OPTForCausalLM
  ccfg = synth.ClipConfig.tiny()
  ocfg = synth.OptConfig.opt_125m()
  opt_dir = os.path.join(tmp, "v/facebook/opt-125m-shape")
  clip_dir = os.path.join(tmp, "v/openai/clip-tiny")
  hf = OPTForCausalLM(OPTConfig(vocab_size=50272, hidden_size=ocfg.hidden_size, num_hidden_layers=ocfg.num_layers,
                                ffn_dim=ocfg.ffn_dim, num_attention_heads=ocfg.num_heads, max_position_embeddings=2048,
                                word_embed_proj_dim=ocfg.hidden_size, do_layer_norm_before=True, dropout=0.0))
  sd_full = {k: v.bfloat16().float() for k, v in synth.opt_state_dict(
    synth.OptConfig(vocab_size=50274, hidden_size=768, num_layers=12, num_heads=12, ffn_dim=3072), seed=5).items()}
  sd_hf = dict(sd_full)
  sd_hf["model.decoder.embed_tokens.weight"] = sd_full["model.decoder.embed_tokens.weight"][:50272].clone()
  sd_hf["lm_head.weight"] = sd_hf["model.decoder.embed_tokens.weight"]
  hf.load_state_dict(sd_hf, strict=True)
  hf.save_pretrained(opt_dir)
  CLIPVisionModel(CLIPVisionConfig(hidden_size=ccfg.hidden_size, intermediate_size=ccfg.intermediate_size,
                                   num_hidden_layers=ccfg.num_layers, num_attention_heads=ccfg.num_heads,
                                   image_size=ccfg.image_size, patch_size=ccfg.patch_size)).save_pretrained(clip_dir)
  ref_utils.get_feature_extractor_for_model = lambda name, **kw: CLIPImageProcessor(
    size={"shortest_edge": ccfg.image_size}, crop_size={"height": ccfg.image_size, "width": ccfg.image_size})
  tok = synth.HashTokenizer()
  args = types.SimpleNamespace(freeze_lm=True, freeze_vm=True, opt_version=opt_dir, visual_encoder=clip_dir,
                               n_visual_tokens=4, ret_emb_dim=256, gen_emb_dim=768, text_emb_layers=[-1],
                               text_fc_mode="gill_mapper", ret_text_fc_mode="linear", num_tokens=8, num_clip_tokens=77,
                               retrieval_token_idx=synth.IMG_TOKEN_IDS, gen_token_idx=synth.IMG_TOKEN_IDS)
  gill = ref_models.GILL(tok, args, load_sd=False, num_gen_images=1)
  gm = gill.model
  clip_sd = {k: v.bfloat16().float() for k, v in synth.clip_state_dict(ccfg, seed=13).items()}
  keys = list(gm.visual_model.state_dict().keys())
  pref = "" if keys[0].startswith("vision_model.") else "vision_model."       # transformers 5.x dropped the prefix
  proj = {}
  synth._linear(proj, "visual_embeddings", 4 * 768, ccfg.hidden_size, 13)
  proj = {k: v.bfloat16().float() for k, v in proj.items()}
  with torch.no_grad():
    gm.visual_model.load_state_dict({k[len(pref):]: v for k, v in clip_sd.items()}, strict=True)
    gm.visual_embeddings.weight.copy_(proj["visual_embeddings.weight"]); gm.visual_embeddings.bias.copy_(proj["visual_embeddings.bias"])
    gm.input_embeddings.weight.copy_(sd_full["model.decoder.embed_tokens.weight"])
    msd = {k: v.bfloat16().float() for k, v in synth.mapper_state_dict(synth.MapperConfig(in_dim=768), seed=7).items()}
    gm.gen_text_hidden_fcs[0].load_state_dict(msd, strict=True)
  gill.eval()
  # ---- F5
  px = synth.normal("golden_pixel_values", (3, 3, ccfg.image_size, ccfg.image_size), 13)
  with torch.no_grad():
    ve = gm.get_visual_embs(px, mode="captioning")
  # ---- F6: a deterministic RGB image larger than the crop, then a text prompt
  rng = np.random.default_rng(13)
  img_arr = rng.integers(0, 256, (48, 64, 3), dtype=np.uint8)
  text = "a small red fox jumps over the lazy dog"
  with torch.no_grad():
    ret = gill.generate_for_images_and_texts([Image.fromarray(img_arr), text], num_words=2, gen_scale_factor=1e5)
  gen = ret[1]["gen"][0]
  # ---- F7: retrieval branch (emb_matrix given): ret_text_hidden_fcs Linear -> normalise -> emb_matrix @ ret_emb.T -> top-3
  # images.  Harness shim: the reference fetches path_array entries over http (utils.get_image_from_url); here the entries
  # are local PNG files, opened by a stand-in with the same resize/convert steps.
  n_img = 24
  img_dir = os.path.join(tmp, "cc3m_stub")
  os.makedirs(img_dir, exist_ok=True)
  paths = []
  for k in range(n_img):
    arr = np.full((20, 20, 3), (7 * k) % 256, dtype=np.uint8)
    arr[:, :, 1] = (13 * k + 5) % 256
    pth = os.path.join(img_dir, f"{k}.png")
    Image.fromarray(arr).save(pth)
    paths.append(pth)
  ref_utils.get_image_from_url = lambda url: Image.open(url).resize((224, 224)).convert("RGB")
  emb_matrix = synth.normal("cc3m_emb_matrix", (n_img, 256), 13)
  emb_matrix = emb_matrix / emb_matrix.norm(dim=-1, keepdim=True)
  rproj = {}
  synth._linear(rproj, "ret_text_hidden_fcs.0.model", 256, 768, 13)
  rproj = {k: v.bfloat16().float() for k, v in rproj.items()}
  with torch.no_grad():
    gm.ret_text_hidden_fcs[0].model.weight.copy_(rproj["ret_text_hidden_fcs.0.model.weight"])
    gm.ret_text_hidden_fcs[0].model.bias.copy_(rproj["ret_text_hidden_fcs.0.model.bias"])
  gill.emb_matrix = emb_matrix
  gill.path_array = paths
  with torch.no_grad():
    ret7 = gill.generate_for_images_and_texts([text], num_words=2, gen_scale_factor=1e5)
  rets = ret7[1]["ret"]
  ret_scores = np.array([r[2] for r in rets], dtype=np.float64)
  ret_ids = np.array([int(np.asarray(r[0])[0, 0, 0]) for r in rets], dtype=np.int64)     # red channel = (7 k) % 256 identifies k
  print("F7 ret", ret_ids.tolist(), ret_scores.tolist(), ret7[1]["decision"])
  # ---- F9: get_visual_embs(mode='retrieval') (gill/models.py:141-146: pooler_output -> visual_fc -> (B,1,256)) and the CLIP
  # rerank of GENERATED images (gill/models.py:724-751).  Harness shim: the reference's sd_pipe is replaced by a stand-in that
  # returns two fixed PIL images (stage 3 has its own fixtures); everything after the call — resize, feature extractor,
  # get_visual_embs(mode='retrieval'), normalise, scores against ret_emb, the sort — is the reference's code.
  vfc = {}
  synth._linear(vfc, "visual_fc", 256, ccfg.hidden_size, 13)
  vfc = {k: v.bfloat16().float() for k, v in vfc.items()}
  with torch.no_grad():
    gm.visual_fc.weight.copy_(vfc["visual_fc.weight"]); gm.visual_fc.bias.copy_(vfc["visual_fc.bias"])
    ve_ret = gm.get_visual_embs(px, mode="retrieval")
  rng9 = np.random.default_rng(17)
  gen_arrs = rng9.integers(0, 256, (2, 64, 64, 3), dtype=np.uint8)
  gen_arrs[0, 0, 0, 0], gen_arrs[1, 0, 0, 0] = 11, 222        # red value of pixel (0,0) identifies the image after the sort

  class _StubPipe:
    def __call__(self, prompt_embeds=None, **kw):
      return types.SimpleNamespace(images=[Image.fromarray(a) for a in gen_arrs[:prompt_embeds.shape[0]]])
  gill.sd_pipe, gill.load_sd, gill.num_gen_images = _StubPipe(), True, 2
  with torch.no_grad():
    ret9 = gill.generate_for_images_and_texts([text], num_words=2, gen_scale_factor=1e5)
  gens = ret9[1]["gen"]
  rerank_scores = np.array([s for _, s in gens], dtype=np.float64)
  rerank_red = np.array([int(np.asarray(im)[0, 0, 0]) for im, _ in gens], dtype=np.int64)
  print("F9", tuple(ve_ret.shape), float(ve_ret.abs().mean()), "rerank", rerank_red.tolist(), rerank_scores.tolist())
  gill.sd_pipe, gill.load_sd, gill.num_gen_images = None, False, 1
  gill.emb_matrix = None
  gill.path_array = None
  np.savez_compressed(os.path.join(OUT, "gill_visual_tiny.npz"), ret_scores=ret_scores, ret_red=ret_ids, n_img=np.int64(n_img),
                      visual_embs_retrieval=ve_ret.numpy(), rerank_images=gen_arrs, rerank_scores=rerank_scores, rerank_red=rerank_red,
                      ret_decision=np.array(str(ret7[1]["decision"])), pixel_values=px.numpy(), visual_embs=ve.numpy(),
                      image=img_arr, text=np.array(text), caption=np.array(ret[0]), decision=np.array(str(ret[1]["decision"])),
                      gen_emb=gen.numpy(), clip_seed=np.int64(13), opt_seed=np.int64(5), mapper_seed=np.int64(7))
  print("F5", tuple(ve.shape), float(ve.abs().mean()), "F6", repr(ret[0]), ret[1]["decision"], tuple(gen.shape))


def _import_reference_sd():
  """gill/custom_sd.py imports diffusers names at module level (custom_sd.py:18-31); diffusers is not installed, so empty
  modules carrying exactly those names are registered first.  None of them contributes behaviour to __call__ except the
  two trivial ones written out here (DiffusionPipeline.device, the output dataclass)."""
  import transformers
  if not hasattr(transformers, "CLIPFeatureExtractor"):
    transformers.CLIPFeatureExtractor = object
  names = ["diffusers", "diffusers.configuration_utils", "diffusers.models", "diffusers.schedulers", "diffusers.utils",
           "diffusers.pipeline_utils", "diffusers.pipelines", "diffusers.pipelines.stable_diffusion",
           "diffusers.pipelines.stable_diffusion.safety_checker"]
  for n in names:
    sys.modules[n] = types.ModuleType(n)
  d = sys.modules
  d["diffusers"].StableDiffusionPipeline = object
  d["diffusers.configuration_utils"].FrozenDict = dict
  d["diffusers.models"].AutoencoderKL = object
  d["diffusers.models"].UNet2DConditionModel = object
  d["diffusers.schedulers"].KarrasDiffusionSchedulers = object
  u = d["diffusers.utils"]
  u.deprecate = lambda *a, **k: None
  u.is_accelerate_available = lambda: False
  u.logging = types.SimpleNamespace(get_logger=lambda name: types.SimpleNamespace(warning=print, info=print))
  u.randn_tensor = lambda shape, generator=None, device=None, dtype=None: torch.randn(shape, generator=generator, dtype=dtype)
  u.replace_example_docstring = lambda doc: (lambda fn: fn)

  class DiffusionPipeline:
    device = torch.device("cpu")
  d["diffusers.pipeline_utils"].DiffusionPipeline = DiffusionPipeline

  class StableDiffusionPipelineOutput:
    def __init__(self, images, nsfw_content_detected):
      self.images, self.nsfw_content_detected = images, nsfw_content_detected
  d["diffusers.pipelines.stable_diffusion"].StableDiffusionPipelineOutput = StableDiffusionPipelineOutput
  d["diffusers.pipelines.stable_diffusion.safety_checker"].StableDiffusionSafetyChecker = object
  if REF not in sys.path:
    sys.path.insert(0, REF)
  sys.modules.pop("gill.custom_sd", None)
  import gill.custom_sd as ref_sd
  return ref_sd


def golden_sd_driver():
  """F8: reference pipeline driver over the oracle's tiny UNet / PNDM scheduler / VAE decoder, 2 prompts, 10 steps."""
  from oracle import scheduler_ref, unet_ref, vae_ref
  ref_sd = _import_reference_sd()
  cfg = synth.UNetConfig.tiny(16)
  vcfg = synth.VAEConfig.tiny(16)
  bfw = lambda sd: {k: v.bfloat16().float() for k, v in sd.items()}  # noqa: E731
  usd, vsd = bfw(synth.unet_state_dict(cfg, seed=21)), bfw(synth.vae_decoder_state_dict(vcfg, seed=22))
  B, steps, guidance = 2, 10, 7.5
  cond = synth.normal("f8_cond", (B, cfg.ctx_len, cfg.cross_attention_dim), 21).bfloat16().float()
  neg = synth.normal("f8_neg", (B, cfg.ctx_len, cfg.cross_attention_dim), 22).bfloat16().float()   # per-sample negatives
  lat0 = synth.initial_latents(B, 4, cfg.sample_size, seed=2024)

  class _Out:
    def __init__(self, **kw):
      self.__dict__.update(kw)

  class UNet:
    config = types.SimpleNamespace(sample_size=cfg.sample_size)
    in_channels = cfg.in_channels
    calls = []

    def __call__(self, x, t, encoder_hidden_states=None, cross_attention_kwargs=None):
      UNet.calls.append((int(t), tuple(x.shape)))
      return _Out(sample=unet_ref.unet_forward(usd, x, torch.full((x.shape[0],), float(t)), encoder_hidden_states,
                                               cfg.block_out_channels, cfg.num_heads, cfg.norm_num_groups))

  class Scheduler(scheduler_ref.PNDMSchedulerRef):
    def set_timesteps(self, n, device=None):
      return super().set_timesteps(n)

    def step(self, model_output, timestep, sample):
      return _Out(prev_sample=super().step(model_output, int(timestep), sample))

  class VAE:
    config = types.SimpleNamespace(scaling_factor=vcfg.scaling_factor, block_out_channels=vcfg.block_out_channels)

    def decode(self, z):   # the pipeline has already divided by 0.18215 (custom_sd.py:387): undo the oracle's own scaling
      return _Out(sample=vae_ref.vae_decode(vsd, z * vcfg.scaling_factor, vcfg.block_out_channels, vcfg.norm_num_groups,
                                            vcfg.scaling_factor))

  pipe = object.__new__(ref_sd.StableDiffusionPipeline)     # __init__ only registers modules / checks configs
  pipe.unet, pipe.scheduler, pipe.vae = UNet(), Scheduler(), VAE()
  pipe.text_encoder = types.SimpleNamespace(dtype=torch.float32)
  pipe.safety_checker, pipe.feature_extractor, pipe.tokenizer = None, None, None
  pipe.vae_scale_factor = 8
  seen = {}
  with torch.no_grad():
    out = pipe(prompt_embeds=cond, negative_prompt_embeds=neg, latents=lat0.clone(), guidance_scale=guidance,
               num_inference_steps=steps, output_type="np", callback=lambda i, t, lat: seen.__setitem__("lat", lat.clone()),
               callback_steps=1)
    # same call with ONE negative embedding for the whole batch is not expressible through the reference's check_inputs
    # (shapes must match), which is how GILL calls it only when negative_prompt_embeds is None; so also record the
    # num_images_per_prompt = 2 expansion order of a single prompt (custom_sd.py:313-316, :361-364)
    seen2 = {}
    pipe.unet.calls.clear()
    out2 = pipe(prompt_embeds=cond[:1], negative_prompt_embeds=neg[:1], latents=lat0.clone(), guidance_scale=guidance,
                num_inference_steps=3, num_images_per_prompt=2, output_type="np",
                callback=lambda i, t, lat: seen2.__setitem__("lat", lat.clone()), callback_steps=1)
  assert out.nsfw_content_detected is None and out.images.shape == (B, 128, 128, 3)
  np.savez_compressed(os.path.join(OUT, "sd_driver_tiny.npz"), cond=cond.numpy(), neg=neg.numpy(), lat0=lat0.numpy(),
                      steps=np.int64(steps), guidance=np.float32(guidance), unet_seed=np.int64(21), vae_seed=np.int64(22),
                      latents=seen["lat"].numpy(), images=out.images.astype(np.float16),
                      latents_n2=seen2["lat"].numpy(), images_n2=out2.images.astype(np.float16),
                      unet_call_timesteps=np.array([c[0] for c in pipe.unet.calls], dtype=np.int64))
  print("F8", tuple(seen["lat"].shape), float(seen["lat"].abs().mean()), tuple(out.images.shape), float(out.images.mean()),
        "n2", tuple(seen2["lat"].shape), [c[0] for c in pipe.unet.calls])


def main():
  os.makedirs(OUT, exist_ok=True)
  if len(sys.argv) > 1 and sys.argv[1] == "sd":   # F8 only (does not need the OPT / CLIP temp models)
    torch.set_num_threads(8)
    golden_sd_driver()
    return
  torch.manual_seed(0)
  torch.set_num_threads(8)
  ref_layers, ref_models = _import_reference()
  golden_mapper(ref_layers)
  tmp = tempfile.mkdtemp(prefix="gill_golden_")
  try:
    golden_gillmodel(ref_models, tmp)
    golden_visual(ref_models, tmp)
    golden_sd_driver()
  finally:
    shutil.rmtree(tmp, ignore_errors=True)
  print("wrote", sorted(os.listdir(OUT)))


if __name__ == "__main__":
  main()
