"""Host-side logic that needs no GPU: sharding, the world_size-2 gather over gloo, tokenizer stand-in, synthetic
inputs, model construction and the reference's error behaviour."""
import os
import sys

import pytest
import torch
import torch.multiprocessing as mp

from gill_amd import parallel, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_bounds_cover_everything():
  for n in (0, 1, 7, 8, 9, 64, 65):
    for w in (1, 2, 3, 8):
      spans = [parallel.shard_bounds(n, r, w) for r in range(w)]
      assert spans[0][0] == 0 and spans[-1][1] == n
      assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
      sizes = [hi - lo for lo, hi in spans]
      assert max(sizes) - min(sizes) <= 1


def _gloo_worker(rank, world, port, n_total, q):
  import torch.distributed as dist
  os.environ["MASTER_ADDR"] = "127.0.0.1"
  os.environ["MASTER_PORT"] = str(port)
  dist.init_process_group("gloo", rank=rank, world_size=world)
  try:
    lo, hi = parallel.shard_range(n_total)
    full = torch.arange(n_total * 4 * 3 * 3, dtype=torch.float32).reshape(n_total, 4, 3, 3)
    local = full[lo:hi] * 1.0                      # this rank's "latents"
    out = parallel.gather_rows(local, n_total)
    q.put((rank, bool(torch.equal(out, full)), tuple(out.shape)))
  finally:
    dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [8, 5, 1])      # 1 < world: rank 1's shard is empty and it must still enter the collective
def test_gather_rows_world2_gloo(n_total):
  ctx = mp.get_context("spawn")
  q = ctx.Queue()
  port = 29500 + (os.getpid() % 2000) + n_total
  procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, n_total, q)) for r in range(2)]
  for p in procs:
    p.start()
  res = [q.get(timeout=120) for _ in procs]
  for p in procs:
    p.join(timeout=60)
  assert all(ok for _, ok, _ in res), res
  assert all(shape == (n_total, 4, 3, 3) for _, _, shape in res)


def test_gather_rows_rejects_none_and_wrong_shard():
  with pytest.raises(ValueError, match="empty shard"):
    parallel.gather_rows(None, 4, distributed=False)
  x = torch.zeros(3, 2)
  assert parallel.gather_rows(x, 3, distributed=False) is x           # single process: identity


def test_prompt_ids_lengths_with_pad_equal_to_eos():
  """load_gill sets pad_token_id = eos_token_id (= bos = 2 for OPT) for tokenizers without a pad token: the length of a tensor
  prompt is the row minus its TRAILING pad run, not a count of non-pad ids (the leading BOS is a pad id too)."""
  g = _tiny_gill()
  g.model.tokenizer.pad_token_id = 2
  ids = torch.tensor([[2, 11, 12, 13, 2, 2], [2, 21, 2, 23, 24, 25], [2, 2, 2, 2, 2, 2]])
  _, lens = g._prompt_ids(ids)
  assert lens.tolist() == [4, 6, 0]
  g.model.tokenizer.pad_token_id = 1
  _, lens = g._prompt_ids(torch.tensor([[2, 11, 12, 1, 1], [2, 21, 22, 23, 24]]))
  assert lens.tolist() == [3, 5]


def test_hash_tokenizer_contract():
  tok = synth.HashTokenizer()
  assert len(tok) == 50274 and tok.cls_token_id == 50265
  ids = tok("a photo of a cat" + "".join(f"[IMG{i}]" for i in range(8)), add_special_tokens=True, return_tensors="pt").input_ids
  assert ids.shape == (1, 1 + 5 + 8) and ids[0, 0] == 2 and ids[0, -8:].tolist() == synth.IMG_TOKEN_IDS
  assert tok("\n", add_special_tokens=False).input_ids == [50118]
  assert tok.batch_decode(ids, skip_special_tokens=True)[0].count("w") == 5


def test_synthetic_inputs_are_deterministic():
  a, b = synth.synthetic_prompt_ids(4, 24, seed=0), synth.synthetic_prompt_ids(4, 24, seed=0)
  assert torch.equal(a, b) and a.shape == (4, 32) and (a[:, 0] == 2).all() and a[:, -8:].tolist() == [synth.IMG_TOKEN_IDS] * 4
  assert a[:, 1:24].min() >= 3 and a[:, 1:24].max() <= 50264
  l1, l2 = synth.initial_latents(2), synth.initial_latents(3)
  assert torch.equal(l1, l2[:2]) and l1.shape == (2, 4, 64, 64)
  w1 = synth.normal("x", (4, 4), 3)
  assert torch.equal(w1, synth.normal("x", (4, 4), 3)) and not torch.equal(w1, synth.normal("y", (4, 4), 3))


def test_unet_state_dict_inventory():
  sd = synth.unet_state_dict(synth.UNetConfig.tiny())
  assert sd["up_blocks.1.resnets.2.conv1.weight"].shape == (256, 256 + 128, 3, 3)
  assert sd["up_blocks.3.resnets.0.conv_shortcut.weight"].shape == (64, 128 + 64, 1, 1)
  assert sd["down_blocks.0.attentions.0.transformer_blocks.0.ff.net.0.proj.weight"].shape == (512, 64)
  assert "down_blocks.3.attentions.0.norm.weight" not in sd and "up_blocks.0.attentions.0.norm.weight" not in sd
  n = sum(v.numel() for v in synth.unet_state_dict(synth.UNetConfig.sd15()).values()) if os.environ.get("GILL_SLOW") else 0
  assert n in (0, 859520964)   # SD-1.5 UNet parameter count


def _tiny_gill(load_sd=False):
  from types import SimpleNamespace
  from gill_amd.models import GILL
  tok = synth.HashTokenizer()
  ocfg = synth.OptConfig(vocab_size=len(tok), hidden_size=128, num_layers=2, num_heads=2, ffn_dim=256, max_positions=128)
  args = SimpleNamespace(freeze_lm=True, freeze_vm=True, opt_version="facebook/opt-tiny-synth",
                         visual_encoder="openai/clip-vit-base-patch16", n_visual_tokens=4, ret_emb_dim=256, gen_emb_dim=768,
                         text_emb_layers=[-1], text_fc_mode="gill_mapper", ret_text_fc_mode="linear", num_tokens=8,
                         num_clip_tokens=77, retrieval_token_idx=synth.IMG_TOKEN_IDS, gen_token_idx=synth.IMG_TOKEN_IDS,
                         opt_state_dict=synth.opt_state_dict(ocfg, seed=1))
  return GILL(tok, args, load_sd=load_sd)


def test_gill_construction_and_reference_error_behaviour():
  g = _tiny_gill()
  keys = g.state_dict().keys()
  # checkpoint key layout of the reference (scripts/prune_model_ckpt.py / models.py:880-893)
  for k in ("model.input_embeddings.weight", "model.lm.model.decoder.embed_tokens.weight",
            "model.gen_text_hidden_fcs.0.fc.weight", "model.gen_text_hidden_fcs.0.tfm.decoder.layers.3.multihead_attn.in_proj_weight",
            "model.ret_text_hidden_fcs.0.model.weight", "model.visual_embeddings.weight", "model.visual_fc.weight",
            "model.logit_scale"):
    assert k in keys, k
  assert g.model.input_embeddings.weight is g.model.lm.model.decoder.embed_tokens.weight
  assert g.model.eval() is None and g.eval() is g                      # reference quirk (models.py:155-161)
  with pytest.raises(NotImplementedError):                              # models.py:629
    g.generate_for_images_and_texts(["x"], num_words=0)
  with pytest.raises(ValueError):                                       # models.py:624
    g.generate_for_images_and_texts([3.14], num_words=2)
  from gill_amd.models import load_gill
  with pytest.raises(ValueError, match="model_args.json"):             # models.py:815-816
    load_gill("/nonexistent")
  with pytest.raises(ValueError, match="empty prompt batch"):
    g.generate_images([])
  # image prompts without CLIP vision weights: a loud NotImplementedError, never a silent skip (and never a CPU fallback)
  from PIL import Image
  with pytest.raises((NotImplementedError, RuntimeError)):
    g.generate_for_images_and_texts([Image.new("RGB", (40, 30)), "x"], num_words=2)


def test_install_as_gill_aliases_the_reference_import_names():
  """`from gill import models` resolves to the MI355X package after gill_amd.install_as_gill()."""
  import subprocess
  import sys
  code = ("import gill_amd; gill_amd.install_as_gill()\n"
          "from gill import models, layers, utils\n"
          "import gill.models as m2\n"
          "assert models is gill_amd.models and m2 is models and layers.TextFcLayer is gill_amd.layers.TextFcLayer\n"
          "assert hasattr(models, 'load_gill') and hasattr(models, 'GILL') and hasattr(models, 'GILLArgs')\n"
          "print('ok')\n")
  import os
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=root, timeout=300)
  assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]
