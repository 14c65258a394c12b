#!/usr/bin/env python
"""Soak / repeatability probe of the full-size hot path: the exact bench.py workload, N consecutive
steps, every stage output (OPT [IMG] hidden states, SD embedding, final latents, decoded image) checked each iteration
for finiteness and compared with iteration 0.  Prints one line per iteration and the first bad index; exit code 1 when
any iteration is non-finite.

  python tools/soak.py --iters 30 [--infer-steps 50] [--prompts 4] [--no-decode] [--small] [--stage unet|all]
Bisect switch (read by libgill_amd): GILL_NO_GRAPH=1 (the tile / N-walk bisect switches of round 2 are gone: profiles/r02_soak_bisect.md)
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def stats(t):
  t = t.float()
  fin = torch.isfinite(t)
  n_bad = int((~fin).sum().item())
  n_nan = int(torch.isnan(t).sum().item())
  amax = float(t[fin].abs().max().item()) if fin.any() else float("nan")
  return n_bad, n_nan, amax


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--iters", type=int, default=30)
  ap.add_argument("--infer-steps", type=int, default=50)
  ap.add_argument("--prompts", type=int, default=4)
  ap.add_argument("--prompt-len", type=int, default=24)
  ap.add_argument("--no-decode", action="store_true")
  ap.add_argument("--small", action="store_true")
  ap.add_argument("--sync-each", action="store_true", help="torch.cuda.synchronize() after every stage")
  ap.add_argument("--sync-mid", action="store_true", help="torch.cuda.synchronize() between the denoise loop and the VAE decode")
  ap.add_argument("--stream", action="store_true", help="run everything on a non-default torch stream")
  ap.add_argument("--fake-vae", type=int, default=0, help="replace the VAE decode by N torch kernels on the current stream")
  a = ap.parse_args()
  dev = torch.device("cuda", 0)
  torch.cuda.set_device(dev)
  from gill_amd import synth
  opt_cfg = synth.OptConfig.opt_125m() if a.small else synth.OptConfig.opt_6_7b()
  unet_cfg = synth.UNetConfig.sd15()
  P = a.prompts
  g = bench.build_model(dev, opt_cfg, unet_cfg, P)
  ids = synth.synthetic_prompt_ids(P, a.prompt_len, seed=0)[:, :a.prompt_len]
  lat0 = synth.initial_latents(P, 4, unet_cfg.sample_size, seed=1337).to(dev)

  rec = {}
  m = g.model
  orig_img = m.img_hidden_states
  mapper = m.gen_text_hidden_fcs[0]
  orig_map = mapper.forward

  def img_hook(*args, **kw):
    raw, emb = orig_img(*args, **kw)
    rec["raw"] = raw.detach().float().clone()
    if a.sync_each:
      torch.cuda.synchronize()
    return raw, emb

  def map_hook(*args, **kw):
    o = orig_map(*args, **kw)
    rec["emb"] = o.detach().float().clone()
    if a.sync_each:
      torch.cuda.synchronize()
    return o
  m.img_hidden_states = img_hook
  mapper.forward = map_hook

  if a.sync_mid:
    cls = g.sd_pipe.__class__
    orig_call = cls.__call__

    def call_sync(self, *args, **kw):
      o = orig_call(self, *args, **kw)
      torch.cuda.synchronize()
      return o
    cls.__call__ = call_sync
  if a.fake_vae:
    cls2 = g.sd_pipe.__class__
    big = torch.zeros(64 << 20, device=dev)

    def fake_decode(self, latents, as_uint8=True, both=False):
      for i in range(a.fake_vae):
        big.add_(1.0)
      return torch.zeros((latents.shape[0], 512, 512, 3), device=dev, dtype=torch.uint8) + 7
    cls2.decode_latents = fake_decode
  if a.stream:
    st = torch.cuda.Stream()
    torch.cuda.set_stream(st)

  first = {}
  bad_at = None
  import time
  for it in range(a.iters):
    t_it = time.perf_counter()
    r = g.generate_images(ids, num_inference_steps=a.infer_steps, guidance_scale=7.5, latents=lat0, decode=not a.no_decode)
    if a.no_decode:
      lat, img = r, None
    else:
      lat, img = r
    torch.cuda.synchronize()
    t_it = time.perf_counter() - t_it
    cur = {"raw": rec["raw"], "emb": rec["emb"], "lat": lat.float().clone()}
    if img is not None:
      cur["img"] = img.float().clone()
    line = [f"it {it:3d} {t_it * 1e3:7.1f} ms"]
    any_bad = False
    for k, v in cur.items():
      n_bad, n_nan, amax = stats(v)
      if it == 0:
        first[k] = v
      rel = float(((v - first[k]).norm() / first[k].norm()).item()) if n_bad == 0 else float("nan")
      line.append(f"{k}: bad={n_bad} nan={n_nan} amax={amax:.4g} rel0={rel:.3e}")
      any_bad |= n_bad > 0
    print(" | ".join(line), flush=True)
    if any_bad and bad_at is None:
      bad_at = it
      # which samples / where
      for k, v in cur.items():
        fin = torch.isfinite(v)
        if not fin.all():
          per = (~fin).reshape(v.shape[0], -1).sum(1).tolist()
          print(f"  first bad iteration {it}: {k} non-finite per sample {per}", flush=True)
  print("FIRST_BAD", bad_at, flush=True)
  sys.exit(1 if bad_at is not None else 0)


if __name__ == "__main__":
  main()
