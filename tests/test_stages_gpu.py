"""Stage-level parity on the MI355X, through the C ABI (ctypes -> libgill_amd.so):

  stage 1+2  OPT [IMG] hidden states + GILLMapper  vs  the REFERENCE outputs in tests/golden (bar: SD-embedding MSE < 1e-4)
  stage 3    UNet forward / CFG+PLMS loop          vs  the CPU oracle (parity unpinned: diffusers is absent, see oracle/)

Weights are bf16-rounded before they reach either side, so the comparison measures the arithmetic, not the storage
rounding the reference's own .bfloat16() model has as well.  Tolerances are stated next to each assert."""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from gill_amd import synth

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _bfw(sd):
  return {k: v.bfloat16().float() for k, v in sd.items()}


def _stats(name, got, ref):
  got, ref = got.float().cpu(), ref.float().cpu()
  mse = ((got - ref) ** 2).mean().item()
  rel = ((got - ref).norm() / ref.norm().clamp_min(1e-12)).item()
  cos = torch.nn.functional.cosine_similarity(got.flatten(), ref.flatten(), dim=0).item()
  print(f"[{name}] mse={mse:.3e} rel_l2={rel:.3e} cos={cos:.6f} max_abs={(got - ref).abs().max().item():.3e} ref_rms={ref.pow(2).mean().sqrt().item():.3f}")
  return mse, rel, cos


# ------------------------------------------------------------------------------------------------ stage 2
@pytest.mark.parametrize("tag,in_dim", [("d768_b2", 768), ("d4096_b1", 4096)])
def test_mapper_vs_reference_golden(cuda, tag, in_dim):
  from gill_amd.layers import TextFcLayer
  g = np.load(os.path.join(GOLD, f"mapper_{tag}.npz"))
  layer = TextFcLayer(in_dim, 768, num_input_tokens=8, num_output_tokens=77, mode="gill_mapper")
  layer.load_state_dict(_bfw(synth.mapper_state_dict(synth.MapperConfig(in_dim=in_dim), seed=int(g["seed"]))), strict=True)
  layer = layer.to(cuda)
  x, e = torch.from_numpy(g["x"]).to(cuda), torch.from_numpy(g["e"]).to(cuda)
  y = layer(x, e)
  mse, rel, cos = _stats(f"mapper {tag}", y, torch.from_numpy(g["y"]))
  assert y.shape == (x.shape[0], 77, 768)
  assert mse < 1e-4          # north_star bar: SD-embedding MSE vs reference < 1e-4
  assert rel < 2e-2
  # broadcasting of input_embs (B vs 1) is numerically the same call
  y2 = layer(x, e.repeat(x.shape[0], 1, 1))
  assert torch.equal(y, y2)


# ------------------------------------------------------------------------------------------------ stage 1 (+2)
def _gill_opt125m(cuda, load_sd=False, sd_pipe=None):
  from gill_amd.models import GILL
  tok = synth.HashTokenizer()
  ocfg = synth.OptConfig(vocab_size=50274, hidden_size=768, num_layers=12, num_heads=12, ffn_dim=3072)
  args = SimpleNamespace(freeze_lm=True, freeze_vm=True, opt_version="facebook/opt-125m", visual_encoder="openai/clip-vit-base-patch16",
                         n_visual_tokens=4, ret_emb_dim=256, gen_emb_dim=768, text_emb_layers=[-1], text_fc_mode="gill_mapper",
                         ret_text_fc_mode="linear", num_tokens=8, num_clip_tokens=77, retrieval_token_idx=synth.IMG_TOKEN_IDS,
                         gen_token_idx=synth.IMG_TOKEN_IDS, opt_state_dict=_bfw(synth.opt_state_dict(ocfg, seed=5)))
  g = GILL(tok, args, load_sd=load_sd, sd_pipe=sd_pipe)
  g.model.gen_text_hidden_fcs[0].load_state_dict(_bfw(synth.mapper_state_dict(synth.MapperConfig(in_dim=768), seed=7)), strict=True)
  g = g.eval()
  g = g.bfloat16()        # the chain load_gill applies (models.py:875-877)
  g = g.cuda()
  return g


@pytest.fixture(scope="module")
def gill125(cuda):
  return _gill_opt125m(cuda)


def test_gillmodel_forward_vs_reference_golden(cuda, gill125):
  """GILLModel.forward(mode='generation') on the ragged right-padded batch of the golden fixture."""
  g = np.load(os.path.join(GOLD, "gillmodel_forward_opt125m.npz"))
  labels, cap = torch.from_numpy(g["labels"]), torch.from_numpy(g["caption_len"])
  out = gill125(torch.zeros(labels.shape[0], 3, 32, 32, device=cuda), labels.to(cuda), cap, mode="generation")
  assert len(out) == 8
  llm_hidden, last_embedding = out[7][0], out[2]
  assert torch.equal(out[1].cpu(), torch.from_numpy(g["full_labels"]))
  mse_h, rel_h, _ = _stats("opt125m [IMG] hidden", llm_hidden, torch.from_numpy(g["llm_hidden"]))
  mse_e, rel_e, _ = _stats("opt125m SD embedding", last_embedding, torch.from_numpy(g["last_embedding"]))
  assert rel_h < 3e-2        # bf16 GEMM operands vs the reference's fp32 CPU run, 12 layers
  assert mse_e < 1e-4        # north_star bar
  assert last_embedding.shape == (labels.shape[0], 77, 768)


def test_generate_loop_and_public_api_vs_reference_golden(cuda, gill125):
  g = np.load(os.path.join(GOLD, "gillmodel_generate_opt125m.npz"))
  prompt = torch.from_numpy(g["prompt"]).to(cuda)
  emb = gill125.model.input_embeddings(prompt)
  ids, embs, logits = gill125.model.generate(emb, 2, gen_scale_factor=1e5)
  assert ids.cpu().tolist() == g["gen_ids"].tolist()                        # [IMG0..7] forced, twice
  T0 = prompt.shape[1]
  _, rel, _ = _stats("generate() hidden at [IMG]", embs[-1][:, T0:T0 + 8], torch.from_numpy(g["hidden_img"]))
  assert rel < 3e-2
  assert logits[0].shape == (1, 50274) and logits[0].device.type == "cpu"   # models.py:471-472
  a = np.load(os.path.join(GOLD, "gill_api_opt125m.npz"))
  ret = gill125.generate_for_images_and_texts([str(a["text"])], num_words=2, gen_scale_factor=1e5)
  assert ret[0] == str(a["caption"]) and str(ret[1]["decision"]) == str(a["decision"]) and ret[1]["ret"] == []
  gen = ret[1]["gen"][0]
  mse, _, _ = _stats("public API gen_emb", gen, torch.from_numpy(a["gen_emb"]))
  assert gen.shape == (1, 77, 768) and mse < 1e-4


def test_generate_kv_cache_matches_full_reforward(cuda, gill125):
  """the KV-cached decode loop against the reference-style full re-forward (which the golden
  test above pins to the reference's own output): same greedy tokens, same hidden states, batch 2, 12 steps."""
  ids = synth.synthetic_prompt_ids(2, 9, seed=11)[:, :9].to(cuda)
  emb = gill125.model.input_embeddings(ids)
  kw = dict(min_word_tokens=12, temperature=0.0)        # [IMG] suppressed: 12 ordinary greedy steps
  out_c, embs_c, logits_c = gill125.model.generate(emb, 12, use_kv_cache=True, **kw)
  out_f, embs_f, logits_f = gill125.model.generate(emb, 12, use_kv_cache=False, **kw)
  assert out_c.shape == (2, 12)
  assert out_c.cpu().tolist() == out_f.cpu().tolist()
  assert len(embs_c) == len(embs_f) == 12 and embs_c[-1].shape == embs_f[-1].shape == (2, 9 + 11, 768)
  _, rel, cos = _stats("kv-cache vs re-forward hidden", embs_c[-1], embs_f[-1])
  assert rel < 2e-2 and cos > 0.999
  fin = torch.isfinite(logits_f[-1])                    # the [IMG] columns are filtered to -inf in both
  assert torch.equal(fin, torch.isfinite(logits_c[-1]))
  _, rel_l, _ = _stats("kv-cache vs re-forward last logits", logits_c[-1][fin], logits_f[-1][fin])
  assert rel_l < 2e-2
  # a forced 8-token [IMG] block goes through the cache as one multi-token step (T_new = 8)
  out_i, embs_i, _ = gill125.model.generate(emb[:1], 3, gen_scale_factor=1e5, use_kv_cache=True)
  out_j, embs_j, _ = gill125.model.generate(emb[:1], 3, gen_scale_factor=1e5, use_kv_cache=False)
  assert out_i.cpu().tolist() == out_j.cpu().tolist() and out_i.shape[1] == 24
  _, rel_i, _ = _stats("kv-cache vs re-forward hidden ([IMG] blocks)", embs_i[-1], embs_j[-1])
  assert rel_i < 2e-2


def test_log_likelihood_scores_vs_oracle(cuda, gill125):
  """GILL.get_log_likelihood_scores (models.py:764-807): <bos> once over two text segments, the LM's mean shifted cross-entropy,
  against the fp32 oracle (opt_ref hidden states -> tied lm_head -> F.cross_entropy).  bf16 GEMM operands over 12 layers against
  fp32: the mean log-probability of ~20 tokens agrees to a few 1e-3 of its value; bar 2e-2 relative."""
  from oracle import opt_ref
  prompts = ["a photo of a small brown dog", "running on the beach at sunset with a red ball"]
  got = gill125.get_log_likelihood_scores(prompts)
  tok = gill125.model.tokenizer
  ids = torch.cat([tok(prompts[0], add_special_tokens=True, return_tensors="pt").input_ids,
                   tok(prompts[1], add_special_tokens=True, return_tensors="pt").input_ids[:, 1:]], dim=1)
  sd = {k: v.float().cpu() for k, v in gill125.model.lm.state_dict().items()}
  emb = opt_ref.opt_embed(sd, ids)
  hid = opt_ref.opt_hidden_states(sd, 12, 12, emb)
  logits = opt_ref.opt_logits(sd, hid)[0]
  want = -torch.nn.functional.cross_entropy(logits[:-1], ids[0, 1:]).item()
  print(f"[log-likelihood] native {got:.5f} oracle {want:.5f}")
  assert abs(got - want) < 2e-2 * abs(want)
  with pytest.raises(ValueError):
    gill125.get_log_likelihood_scores([3])


def test_image_prompt_vs_reference_golden(cuda):
  """CLIP vision tower + visual_embeddings (get_visual_embs) and the public API with a PIL image
  in the prompt list, against the reference's own outputs (tests/golden/gill_visual_tiny.npz, oracle/gen_golden.py F5-F6)."""
  from PIL import Image
  from gill_amd.models import GILL
  g = np.load(os.path.join(GOLD, "gill_visual_tiny.npz"))
  ccfg = synth.ClipConfig.tiny()
  tok = synth.HashTokenizer()
  ocfg = synth.OptConfig(vocab_size=50274, hidden_size=768, num_layers=12, num_heads=12, ffn_dim=3072)
  args = SimpleNamespace(freeze_lm=True, freeze_vm=True, opt_version="facebook/opt-125m", visual_encoder="openai/clip-tiny",
                         n_visual_tokens=4, ret_emb_dim=256, gen_emb_dim=768, text_emb_layers=[-1], text_fc_mode="gill_mapper",
                         ret_text_fc_mode="linear", num_tokens=8, num_clip_tokens=77, retrieval_token_idx=synth.IMG_TOKEN_IDS,
                         gen_token_idx=synth.IMG_TOKEN_IDS, opt_state_dict=_bfw(synth.opt_state_dict(ocfg, seed=int(g["opt_seed"]))),
                         clip_state_dict=_bfw(synth.clip_state_dict(ccfg, seed=int(g["clip_seed"]))), clip_config=ccfg)
  m = GILL(tok, args, load_sd=False)
  proj = {}
  synth._linear(proj, "visual_embeddings", 4 * 768, ccfg.hidden_size, int(g["clip_seed"]))
  with torch.no_grad():
    m.model.visual_embeddings.weight.copy_(proj["visual_embeddings.weight"].bfloat16().float())
    m.model.visual_embeddings.bias.copy_(proj["visual_embeddings.bias"].bfloat16().float())
  m.model.gen_text_hidden_fcs[0].load_state_dict(_bfw(synth.mapper_state_dict(synth.MapperConfig(in_dim=768), seed=int(g["mapper_seed"]))),
                                                 strict=True)
  m = m.eval().bfloat16().cuda()
  ve = m.model.get_visual_embs(torch.from_numpy(g["pixel_values"]).to(cuda), mode="captioning")
  assert ve.shape == (3, 4, 768)
  mse, rel, cos = _stats("get_visual_embs", ve, torch.from_numpy(g["visual_embs"]))
  assert rel < 3e-2 and cos > 0.999
  ret = m.generate_for_images_and_texts([Image.fromarray(g["image"]), str(g["text"])], num_words=2, gen_scale_factor=1e5)
  assert ret[0] == str(g["caption"]) and str(ret[1]["decision"]) == str(g["decision"])
  gen = ret[1]["gen"][0]
  mse, _, _ = _stats("public API gen_emb with an image prompt", gen, torch.from_numpy(g["gen_emb"]))
  assert gen.shape == (1, 77, 768) and mse < 1e-4


def test_retrieval_visual_embs_and_rerank_vs_reference_golden(cuda, tmp_path):
  """Golden F9: GILLModel.get_visual_embs(mode='retrieval') (gill/models.py:141-146: CLIP pooler_output -> visual_fc -> (B,1,256))
  and the CLIP rerank of generated images (gill/models.py:733-751: resize, feature extractor, retrieval embeddings, normalise,
  scores against ret_emb, sort) against the REFERENCE's own outputs.  As in the fixture's generating script, the SD pipeline is a
  stand-in that returns the fixture's two fixed images: stage 3 has its own fixtures, the code under test is everything after."""
  from PIL import Image
  from gill_amd.models import GILL
  g = np.load(os.path.join(GOLD, "gill_visual_tiny.npz"))
  ccfg = synth.ClipConfig.tiny()
  tok = synth.HashTokenizer()
  ocfg = synth.OptConfig(vocab_size=50274, hidden_size=768, num_layers=12, num_heads=12, ffn_dim=3072)
  args = SimpleNamespace(freeze_lm=True, freeze_vm=True, opt_version="facebook/opt-125m", visual_encoder="openai/clip-tiny",
                         n_visual_tokens=4, ret_emb_dim=256, gen_emb_dim=768, text_emb_layers=[-1], text_fc_mode="gill_mapper",
                         ret_text_fc_mode="linear", num_tokens=8, num_clip_tokens=77, retrieval_token_idx=synth.IMG_TOKEN_IDS,
                         gen_token_idx=synth.IMG_TOKEN_IDS, opt_state_dict=_bfw(synth.opt_state_dict(ocfg, seed=int(g["opt_seed"]))),
                         clip_state_dict=_bfw(synth.clip_state_dict(ccfg, seed=int(g["clip_seed"]))), clip_config=ccfg)
  n_img = int(g["n_img"])
  paths = []
  for k in range(n_img):
    arr = np.full((20, 20, 3), (7 * k) % 256, dtype=np.uint8)
    arr[:, :, 1] = (13 * k + 5) % 256
    p = str(tmp_path / f"{k}.png")
    Image.fromarray(arr).save(p)
    paths.append(p)
  emb_matrix = synth.normal("cc3m_emb_matrix", (n_img, 256), int(g["clip_seed"]))
  emb_matrix = emb_matrix / emb_matrix.norm(dim=-1, keepdim=True)
  gen_arrs = g["rerank_images"]

  class _StubPipe:      # the stand-in of oracle/gen_golden.py F9
    _vae = True         # "holds VAE weights": .images are PIL images (gill/custom_sd.py:491 default output_type)

    def __call__(self, prompt_embeds=None, **kw):
      return SimpleNamespace(images=[Image.fromarray(a) for a in gen_arrs[:prompt_embeds.shape[0]]])
  m = GILL(tok, args, path_array=paths, emb_matrix=emb_matrix, load_sd=True, sd_pipe=_StubPipe(), num_gen_images=2)
  w = {}
  synth._linear(w, "visual_fc", 256, ccfg.hidden_size, int(g["clip_seed"]))
  synth._linear(w, "ret_text_hidden_fcs.0.model", 256, 768, int(g["clip_seed"]))
  with torch.no_grad():
    m.model.visual_fc.weight.copy_(w["visual_fc.weight"].bfloat16().float())
    m.model.visual_fc.bias.copy_(w["visual_fc.bias"].bfloat16().float())
    m.model.ret_text_hidden_fcs[0].model.weight.copy_(w["ret_text_hidden_fcs.0.model.weight"].bfloat16().float())
    m.model.ret_text_hidden_fcs[0].model.bias.copy_(w["ret_text_hidden_fcs.0.model.bias"].bfloat16().float())
  m.model.gen_text_hidden_fcs[0].load_state_dict(_bfw(synth.mapper_state_dict(synth.MapperConfig(in_dim=768), seed=int(g["mapper_seed"]))),
                                                 strict=True)
  m = m.eval().bfloat16().cuda()
  m.emb_matrix = emb_matrix.to(cuda)
  ve = m.model.get_visual_embs(torch.from_numpy(g["pixel_values"]).to(cuda), mode="retrieval")
  assert ve.shape == (3, 1, 256)
  _, rel, cos = _stats("get_visual_embs(mode='retrieval') vs the reference", ve, torch.from_numpy(g["visual_embs_retrieval"]))
  assert rel < 3e-2 and cos > 0.999
  ret = m.generate_for_images_and_texts([str(g["text"])], num_words=2, gen_scale_factor=1e5)
  gens = ret[1]["gen"]
  reds = [int(np.asarray(im)[0, 0, 0]) for im, _ in gens]
  scores = np.array([sc for _, sc in gens])
  print("[rerank] order", reds, "scores", scores.tolist(), "reference", g["rerank_red"].tolist(), g["rerank_scores"].tolist())
  assert reds == g["rerank_red"].tolist()
  assert np.abs(scores - g["rerank_scores"]).max() < 3e-3


def test_retrieval_branch_vs_reference_golden(cuda, tmp_path):
  """ret_text_hidden_fcs Linear -> normalise -> emb_matrix @ ret_emb.T -> top-3 local images,
  against the reference's own scores / picks (golden F7), plus the decision MLP against plain fp32 math."""
  from PIL import Image
  from gill_amd.models import GILL
  g = np.load(os.path.join(GOLD, "gill_visual_tiny.npz"))
  tok = synth.HashTokenizer()
  ocfg = synth.OptConfig(vocab_size=50274, hidden_size=768, num_layers=12, num_heads=12, ffn_dim=3072)
  args = SimpleNamespace(freeze_lm=True, freeze_vm=True, opt_version="facebook/opt-125m", visual_encoder="openai/clip-vit-base-patch16",
                         n_visual_tokens=4, ret_emb_dim=256, gen_emb_dim=768, text_emb_layers=[-1], text_fc_mode="gill_mapper",
                         ret_text_fc_mode="linear", num_tokens=8, num_clip_tokens=77, retrieval_token_idx=synth.IMG_TOKEN_IDS,
                         gen_token_idx=synth.IMG_TOKEN_IDS, opt_state_dict=_bfw(synth.opt_state_dict(ocfg, seed=int(g["opt_seed"]))))
  n_img = int(g["n_img"])
  paths = []
  for k in range(n_img):
    arr = np.full((20, 20, 3), (7 * k) % 256, dtype=np.uint8)
    arr[:, :, 1] = (13 * k + 5) % 256
    p = str(tmp_path / f"{k}.png")
    Image.fromarray(arr).save(p)
    paths.append(p)
  emb_matrix = synth.normal("cc3m_emb_matrix", (n_img, 256), int(g["clip_seed"]))
  emb_matrix = emb_matrix / emb_matrix.norm(dim=-1, keepdim=True)
  m = GILL(tok, args, path_array=paths, emb_matrix=emb_matrix, load_sd=False)
  rproj = {}
  synth._linear(rproj, "ret_text_hidden_fcs.0.model", 256, 768, int(g["clip_seed"]))
  with torch.no_grad():
    m.model.ret_text_hidden_fcs[0].model.weight.copy_(rproj["ret_text_hidden_fcs.0.model.weight"].bfloat16().float())
    m.model.ret_text_hidden_fcs[0].model.bias.copy_(rproj["ret_text_hidden_fcs.0.model.bias"].bfloat16().float())
  m.model.gen_text_hidden_fcs[0].load_state_dict(_bfw(synth.mapper_state_dict(synth.MapperConfig(in_dim=768), seed=int(g["mapper_seed"]))),
                                                 strict=True)
  dec = torch.nn.Sequential(torch.nn.Dropout(0.5), torch.nn.Linear(768, 2))
  with torch.no_grad():
    dec[1].weight.copy_(synth.normal("decision.w", (2, 768), 3, 0.05).bfloat16().float())
    dec[1].bias.copy_(synth.normal("decision.b", (2,), 3, 0.1))
  m.decision_model = dec.eval()
  m = m.eval().bfloat16().cuda()
  m.emb_matrix = emb_matrix.to(cuda)
  ret = m.generate_for_images_and_texts([str(g["text"])], num_words=2, gen_scale_factor=1e5)
  rets = ret[1]["ret"]
  reds = [int(np.asarray(r[0])[0, 0, 0]) for r in rets]
  scores = np.array([r[2] for r in rets])
  print("[retrieval] picks", reds, "scores", scores.tolist(), "reference", g["ret_red"].tolist(), g["ret_scores"].tolist())
  assert reds == g["ret_red"].tolist() and all(r[1] == "ret" and r[0].size == (224, 224) for r in rets)
  assert np.abs(scores - g["ret_scores"]).max() < 3e-3
  # decision head: softmax(Linear(raw_emb[:, 0])) of the first [IMG] hidden state
  prompt = tok(str(g["text"]), add_special_tokens=True, return_tensors="pt").input_ids.to(cuda)
  _, embs, _ = m.model.generate(m.model.input_embeddings(prompt), 2, gen_scale_factor=1e5)
  h0 = embs[-1][:, prompt.shape[1], :].float().cpu()
  logits = h0 @ dec[1].weight.float().cpu().T + dec[1].bias.float().cpu()
  exp = logits.softmax(-1)[0].tolist()
  got = ret[1]["decision"]
  print("[decision]", got, "expected", exp)
  assert got[0] == {0: "gen", 1: "ret"}[int(logits.argmax())] and np.abs(np.array(got[1]) - np.array(exp)).max() < 2e-2
  assert ret[1]["gen"][0].shape == (1, 77, 768)


def test_public_api_all_branches_with_sd(cuda, tmp_path):
  """generate_for_images_and_texts with everything switched on (small models): retrieval, decision, Stable Diffusion with
  VAE decode to PIL, and the CLIP rerank of the generated images (models.py:724-751)."""
  from PIL import Image
  from gill_amd.models import GILL
  from gill_amd.sd import GillSDPipeline
  ucfg = synth.UNetConfig(block_out_channels=(64, 128, 256, 256), num_heads=4, cross_attention_dim=768, sample_size=16)
  uncond = synth.uncond_context(ucfg.ctx_len, ucfg.cross_attention_dim, seed=3).bfloat16().float()
  vcfg = synth.VAEConfig.tiny(16)
  pipe = GillSDPipeline(_bfw(synth.unet_state_dict(ucfg, seed=3)), ucfg, uncond, cuda, max_batch=8,
                        vae_state=_bfw(synth.vae_decoder_state_dict(vcfg, seed=5)), vae_cfg=vcfg)
  ccfg = synth.ClipConfig.tiny()
  tok = synth.HashTokenizer()
  ocfg = synth.OptConfig(vocab_size=50274, hidden_size=768, num_layers=12, num_heads=12, ffn_dim=3072)
  args = SimpleNamespace(freeze_lm=True, freeze_vm=True, opt_version="facebook/opt-125m", visual_encoder="openai/clip-tiny",
                         n_visual_tokens=4, ret_emb_dim=256, gen_emb_dim=768, text_emb_layers=[-1], text_fc_mode="gill_mapper",
                         ret_text_fc_mode="linear", num_tokens=8, num_clip_tokens=77, retrieval_token_idx=synth.IMG_TOKEN_IDS,
                         gen_token_idx=synth.IMG_TOKEN_IDS, opt_state_dict=_bfw(synth.opt_state_dict(ocfg, seed=5)),
                         clip_state_dict=_bfw(synth.clip_state_dict(ccfg, seed=13)), clip_config=ccfg)
  paths = []
  for k in range(8):
    p = str(tmp_path / f"{k}.png")
    Image.fromarray(np.full((12, 12, 3), 20 * k, dtype=np.uint8)).save(p)
    paths.append(p)
  emb_matrix = synth.normal("api_emb_matrix", (8, 256), 2)
  emb_matrix = emb_matrix / emb_matrix.norm(dim=-1, keepdim=True)
  m = GILL(tok, args, path_array=paths, emb_matrix=emb_matrix, load_sd=True, sd_pipe=pipe, num_gen_images=2)
  m.model.gen_text_hidden_fcs[0].load_state_dict(_bfw(synth.mapper_state_dict(synth.MapperConfig(in_dim=768), seed=7)), strict=True)
  m = m.eval().bfloat16().cuda()
  m.emb_matrix = emb_matrix.to(cuda)
  text = "two birds on a wire"
  ret = m.generate_for_images_and_texts([text], num_words=2, gen_scale_factor=1e5, num_inference_steps=4,
                                        generator=torch.Generator("cpu").manual_seed(5))
  out = ret[1]
  assert len(out["ret"]) == 3 and len(out["gen"]) == 2
  (im0, s0), (im1, s1) = out["gen"]
  assert im0.size == (128, 128) and im1.size == (128, 128) and np.isfinite([s0, s1]).all() and s0 >= s1
  # the rerank score of the best image, recomputed: cos(visual_fc(CLIP(image)), ret_text_hidden_fcs(raw_emb)[0])
  prompt = tok(text, add_special_tokens=True, return_tensors="pt").input_ids.to(cuda)
  _, embs, _ = m.model.generate(m.model.input_embeddings(prompt), 2, gen_scale_factor=1e5)
  raw = embs[-1][:, prompt.shape[1]:prompt.shape[1] + 8, :]
  r = m.model.ret_text_hidden_fcs[0](raw, None)[:, 0, :].float()
  r = r / r.norm(dim=-1, keepdim=True)
  from gill_amd import utils
  px = utils.get_pixel_values_for_model(m.model.feature_extractor, im0.resize((224, 224)).convert("RGB"))[None].to(cuda)
  v = m.model.get_visual_embs(px, mode="retrieval").float().reshape(1, -1)
  v = v / v.norm(dim=-1, keepdim=True)
  print("[rerank] scores", s0, s1, "recomputed best", float((v * r).sum()))
  assert abs(float((v * r).sum()) - s0) < 2e-2


@pytest.mark.skipif(os.environ.get("GILL_SKIP_SLOW") == "1", reason="slow CPU oracle")
def test_clip_vit_l14_vs_oracle(cuda):
  """The full ViT-L/14 geometry the reference uses (224 px, 14 px patches -> K = 588 padded to 640, 257 tokens, 24 layers)."""
  import ctypes as C
  from gill_amd import _native as N
  from oracle import clip_ref
  cfg = synth.ClipConfig.vit_l14()
  sd = _bfw(synth.clip_state_dict(cfg, seed=21))
  px = synth.normal("vitl_px", (2, 3, 224, 224), 21)
  ref = clip_ref.clip_pooler_output(sd, px, cfg.patch_size, cfg.num_heads)
  c = N.gill_clip_config(image_size=cfg.image_size, patch_size=cfg.patch_size, hidden_size=cfg.hidden_size, num_layers=cfg.num_layers,
                         num_heads=cfg.num_heads, intermediate_size=cfg.intermediate_size, max_batch=2)
  arr, keep = N.make_tensor_table(sd, cuda)
  h = C.c_void_p()
  N.check(N.lib().gill_clip_create(C.byref(h), C.byref(c), arr, len(keep)))
  del keep
  out = torch.empty((2, cfg.hidden_size), device=cuda, dtype=torch.float32)
  N.check(N.lib().gill_clip_forward(h, N.ptr(px.to(cuda)), 2, N.ptr(out), N.current_stream()))
  torch.cuda.synchronize()
  N.lib().gill_clip_destroy(h)
  _, rel, cos = _stats("CLIP ViT-L/14 pooler_output", out, ref)
  assert rel < 3e-2 and cos > 0.999


def test_opt_forward_vs_oracle_causal_lengths(cuda, gill125):
  """Single OPT pass at several sequence lengths (kv tile edges 31/32/33, 64, 100) against the oracle."""
  from oracle import opt_ref
  cfg = gill125.model.opt_cfg
  sd = {k: v.float().cpu() for k, v in gill125.model.lm.state_dict().items()}
  for T in (5, 31, 32, 33, 64, 100):
    ids = synth.synthetic_prompt_ids(2, T, seed=T)[:, :T]
    emb = opt_ref.opt_embed(sd, ids)
    ref = opt_ref.opt_hidden_states(sd, cfg.num_layers, cfg.num_heads, emb)
    got = gill125.model._lm_forward_hidden(emb.to(cuda))
    _, rel, _ = _stats(f"opt forward T={T}", got, ref)
    assert rel < 3e-2


# ------------------------------------------------------------------------------------------------ stage 3
def _tiny_pipe(cuda, sample_size=16, max_batch=8):
  from gill_amd.sd import GillSDPipeline
  cfg = synth.UNetConfig.tiny(sample_size)
  sd = _bfw(synth.unet_state_dict(cfg, seed=3))
  uncond = synth.uncond_context(cfg.ctx_len, cfg.cross_attention_dim, seed=3).bfloat16().float()
  return cfg, sd, uncond, GillSDPipeline(sd, cfg, uncond, cuda, max_batch=max_batch)


def test_unet_forward_tiny_vs_oracle(cuda):
  from oracle import unet_ref
  cfg, sd, _, pipe = _tiny_pipe(cuda)
  B = 3
  x = synth.normal("unet_x", (B, 4, 16, 16), 9)
  ctx = synth.normal("unet_ctx", (B, 77, cfg.cross_attention_dim), 9).bfloat16().float()
  t = torch.tensor([981.0, 501.0, 1.0])
  ref = unet_ref.unet_forward(sd, x, t, ctx, cfg.block_out_channels, cfg.num_heads, cfg.norm_num_groups)
  got = pipe.unet(x, t, ctx)
  mse, rel, cos = _stats("unet tiny forward", got, ref)
  assert got.shape == ref.shape
  assert rel < 5e-2 and cos > 0.998     # ~60 bf16 layers against an fp32 oracle


def test_denoise_tiny_vs_oracle(cuda):
  """Whole CFG + PLMS loop (10 steps = 11 UNet calls of batch 2B) incl. the repeated warm-up step."""
  from oracle import pipeline_ref
  cfg, sd, uncond, pipe = _tiny_pipe(cuda)
  B = 2
  cond = synth.normal("dn_cond", (B, 77, cfg.cross_attention_dim), 4).bfloat16().float()
  lat0 = synth.initial_latents(B, 4, 16, seed=1337)
  ref = pipeline_ref.denoise(sd, cond, uncond, lat0, 10, 7.5, cfg.block_out_channels, cfg.num_heads, cfg.norm_num_groups)
  got = pipe(prompt_embeds=cond, latents=lat0, guidance_scale=7.5, num_inference_steps=10).images
  mse, rel, cos = _stats("denoise tiny 10 steps", got, ref)
  assert rel < 4e-2 and cos > 0.995     # 11 recurrent bf16 UNet calls; guidance 7.5 amplifies eps differences
  # no-CFG path (guidance <= 1: custom_sd.py:588)
  ref1 = pipeline_ref.denoise(sd, cond, None, lat0, 5, 1.0, cfg.block_out_channels, cfg.num_heads, cfg.norm_num_groups)
  got1 = pipe(prompt_embeds=cond, latents=lat0, guidance_scale=1.0, num_inference_steps=5).images
  _, rel1, cos1 = _stats("denoise tiny no-CFG 5 steps", got1, ref1)
  assert rel1 < 5e-2 and cos1 > 0.998
  # batch invariance: prompts are independent end to end (what multi-GPU sharding relies on)
  got_b0 = pipe(prompt_embeds=cond[:1], latents=lat0[:1], guidance_scale=7.5, num_inference_steps=10).images
  _, relb, _ = _stats("batch invariance", got_b0, got[:1])
  # not bitwise: the split-K factor (hence the fp32 summation order) depends on the batch; 11 recurrent steps at
  # guidance 7.5 amplify that rounding noise to the same level as the distance to the fp32 oracle above
  assert relb < 2e-2


def test_sd2_geometry_tiny_vs_oracle(cuda):
  """BASELINE.json configs[3] (SD-2.1-768 UNet) as a parity case, reduced width: a different head count at every level
  (fixed head dim 64), and v-prediction in the PLMS update."""
  from gill_amd.sd import GillSDPipeline
  from oracle import pipeline_ref, unet_ref
  cfg = synth.UNetConfig.tiny_sd2(16)
  sd = _bfw(synth.unet_state_dict(cfg, seed=4))
  uncond = synth.uncond_context(cfg.ctx_len, cfg.cross_attention_dim, seed=4).bfloat16().float()
  pipe = GillSDPipeline(sd, cfg, uncond, cuda, max_batch=8)
  B = 2
  x = synth.normal("sd2_x", (B, 4, 16, 16), 9)
  ctx = synth.normal("sd2_ctx", (B, 77, cfg.cross_attention_dim), 9).bfloat16().float()
  t = torch.tensor([981.0, 301.0])
  ref = unet_ref.unet_forward(sd, x, t, ctx, cfg.block_out_channels, cfg.heads_per_level, cfg.norm_num_groups)
  got = pipe.unet(x, t, ctx)
  _, rel, cos = _stats("SD-2.x geometry UNet forward", got, ref)
  assert rel < 5e-2 and cos > 0.998
  cond = ctx[:1]
  lat0 = synth.initial_latents(1, 4, 16, seed=1337)
  refl = pipeline_ref.denoise(sd, cond, uncond, lat0, 8, 7.5, cfg.block_out_channels, cfg.heads_per_level, cfg.norm_num_groups,
                              prediction_type="v_prediction")
  gotl = pipe(prompt_embeds=cond, latents=lat0, guidance_scale=7.5, num_inference_steps=8).images
  _, rell, cosl = _stats("SD-2.x v-prediction denoise 8 steps", gotl, refl)
  # v-prediction feeds sqrt(1 - a_t) * sample back through the update: bf16 differences compound faster than with epsilon
  assert rell < 1.2e-1 and cosl > 0.99


@pytest.mark.skipif(os.environ.get("GILL_SKIP_SLOW") == "1", reason="slow CPU oracle")
def test_unet_forward_sd15_vs_oracle(cuda):
  """Full-size SD-1.5 UNet (860 M parameters, 64x64 latents), one forward of batch 2 against the CPU oracle."""
  from gill_amd.sd import GillSDPipeline
  from oracle import unet_ref
  cfg = synth.UNetConfig.sd15()
  sd = _bfw(synth.unet_state_dict(cfg, seed=0))
  uncond = synth.uncond_context(seed=0)
  pipe = GillSDPipeline(sd, cfg, uncond, cuda, max_batch=2)
  x = synth.initial_latents(2, 4, 64, seed=1337)
  ctx = torch.cat([uncond, synth.normal("sd15_ctx", (1, 77, 768), 2)], 0).bfloat16().float()
  t = torch.tensor([961.0, 961.0])
  ref = unet_ref.unet_forward(sd, x, t, ctx)
  got = pipe.unet(x, t, ctx)
  mse, rel, cos = _stats("unet SD-1.5 forward", got, ref)
  assert rel < 5e-2 and cos > 0.998


def test_generate_images_end_to_end_tiny(cuda):
  """NEW batched entry: token ids -> OPT -> mapper -> UNet loop, against the composed oracle."""
  from oracle import pipeline_ref
  cfg, usd, uncond, pipe = _tiny_pipe(cuda)
  # the tiny UNet takes a 128-wide context: give the mapper a 128-d output via gen_emb_dim... the shipped mapper
  # hard-codes 768 (layers.py:52), so run stages 1-2 at 768 and check stage 3 separately on its own embedding.
  g = _gill_opt125m(cuda, load_sd=False)
  ids = synth.synthetic_prompt_ids(4, 12, seed=8)[:, :12]
  embs = g.generate_images(ids, distributed=False)
  osd = {k: v.float().cpu() for k, v in g.model.lm.state_dict().items()}
  msd = {k: v.float().cpu() for k, v in g.model.gen_text_hidden_fcs[0].state_dict().items()}
  full = torch.cat([ids, torch.tensor([synth.IMG_TOKEN_IDS] * 4)], 1)
  ref = pipeline_ref.sd_embedding(osd, msd, 12, 12, full, torch.full((4,), full.shape[1] - 1))
  mse, _, _ = _stats("generate_images SD embedding (B=4)", embs, ref)
  assert embs.shape == (4, 77, 768) and mse < 1e-4


# ------------------------------------------------------------------------------------------------ stage 3b (VAE decode)
def test_vae_decode_tiny_vs_oracle(cuda):
  """custom_sd.py:385-392 + numpy_to_pil: latents -> vae.decode(latents / 0.18215).sample -> uint8 HWC."""
  from oracle import vae_ref
  cfg, sd, uncond, pipe = _tiny_pipe(cuda)
  vcfg = synth.VAEConfig.tiny(16)
  vsd = _bfw(synth.vae_decoder_state_dict(vcfg, seed=5))
  pipe.load_vae(vsd, vcfg)
  B = 3
  lat = synth.initial_latents(B, 4, 16, seed=77) * 0.18215 * 3.0   # decoder input of a few units, as real latents are
  ref = vae_ref.vae_decode(vsd, lat, vcfg.block_out_channels, vcfg.norm_num_groups, vcfg.scaling_factor)
  got = pipe.decode_latents(lat, as_uint8=False)
  assert got.shape == ref.shape == (B, 3, 128, 128)
  mse, rel, cos = _stats("vae decode tiny", got, ref)
  assert rel < 4e-2 and cos > 0.999
  f32, u8 = pipe.decode_latents(lat, both=True)
  u8 = u8.cpu()
  assert u8.shape == (B, 128, 128, 3) and u8.dtype == torch.uint8
  # the uint8 conversion is exact on the fp32 output of the same decode ...
  assert torch.equal(u8, vae_ref.to_uint8(f32.cpu()))
  # ...and within a few grey levels of the fp32 oracle's image
  d = (u8.int() - vae_ref.to_uint8(ref).int()).abs()
  print(f"[vae uint8] max level diff {d.max().item()}, mean {d.float().mean().item():.3f}")
  assert d.float().mean().item() < 1.5
  # output_type plumbing of the pipeline object
  cond = synth.normal("dn_cond", (1, 77, cfg.cross_attention_dim), 4).bfloat16().float()
  ims = pipe(prompt_embeds=cond, latents=lat[:1], guidance_scale=7.5, num_inference_steps=3, output_type="pil").images
  assert len(ims) == 1 and ims[0].size == (128, 128)
  # the default output type is the reference's (custom_sd.py:491 output_type="pil") once VAE weights are loaded
  dflt = pipe(prompt_embeds=cond, latents=lat[:1], guidance_scale=7.5, num_inference_steps=3).images
  assert isinstance(dflt, list) and np.array_equal(np.asarray(dflt[0]), np.asarray(ims[0]))
  arr = pipe(prompt_embeds=cond, latents=lat[:1], guidance_scale=7.5, num_inference_steps=3, output_type="np").images
  assert arr.shape == (1, 128, 128, 3) and 0.0 <= arr.min() and arr.max() <= 1.0
  # run-to-run: every reduction is fixed-order (per-slab partial sums, no atomics), so two runs are bit-identical
  dd = np.abs((arr[0] * 255).round().astype(np.int32) - np.asarray(ims[0]).astype(np.int32))
  assert dd.max() == 0, f"denoise + decode differs run to run: max level diff {dd.max()}"
  u8b = pipe.decode_latents(lat, as_uint8=True).cpu()
  assert torch.equal(u8b, u8), "decode differs run to run"


@pytest.mark.skipif(os.environ.get("GILL_SKIP_SLOW") == "1", reason="slow CPU oracle")
def test_vae_decode_sd15_vs_oracle(cuda):
  """Full SD-1.5 decoder geometry (64x64 latents -> 512x512), batch 1, against the fp32 oracle."""
  from oracle import vae_ref
  from gill_amd.sd import GillSDPipeline
  ucfg = synth.UNetConfig.tiny(64)
  pipe = GillSDPipeline(_bfw(synth.unet_state_dict(ucfg, seed=3)), ucfg,
                        synth.uncond_context(ucfg.ctx_len, ucfg.cross_attention_dim, seed=3), cuda, max_batch=2)
  vcfg = synth.VAEConfig.sd15()
  vsd = _bfw(synth.vae_decoder_state_dict(vcfg, seed=6))
  pipe.load_vae(vsd, vcfg)
  lat = synth.initial_latents(1, 4, 64, seed=78) * 0.18215 * 3.0
  ref = vae_ref.vae_decode(vsd, lat, vcfg.block_out_channels, vcfg.norm_num_groups, vcfg.scaling_factor)
  got = pipe.decode_latents(lat, as_uint8=False)
  mse, rel, cos = _stats("vae decode sd1.5", got, ref)
  assert rel < 4e-2 and cos > 0.999
