#!/usr/bin/env python3
"""Does running two denoise loops side by side (two UNet handles, each on its own stream) hide the per-kernel fixed costs?
Times, for the SD-1.5 UNet at 50 steps:  one handle with B prompts  vs  two handles with B/2 prompts each, concurrently.
  python tools/two_stream_probe.py"""
import os
import sys
import time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gill_amd import synth
from gill_amd.sd import GillSDPipeline

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
cfg = synth.UNetConfig.sd15()
unet_sd = bench.gpu_state_dict(lambda c, meta: bench.shapes_of("unet_state_dict", c), cfg, dev, 1)
uncond = synth.uncond_context(cfg.ctx_len, cfg.cross_attention_dim, 0)
pipes = [GillSDPipeline(unet_sd, cfg, uncond, dev, max_batch=16) for _ in range(2)]
streams = [torch.cuda.Stream(dev) for _ in range(2)]
steps = 50


def run(assign):          # assign: list of (pipe index, prompts)
  embs = [torch.randn(n, cfg.ctx_len, cfg.cross_attention_dim, device=dev).bfloat16() for _, n in assign]
  lats = [synth.initial_latents(n, 4, cfg.sample_size, seed=7 + i).to(dev) for i, (_, n) in enumerate(assign)]
  torch.cuda.synchronize()

  def once():
    for (pi, n), e, l in zip(assign, embs, lats):
      with torch.cuda.stream(streams[pi]):
        pipes[pi](prompt_embeds=e, latents=l, num_inference_steps=steps, guidance_scale=7.5, output_type="latent")
  once(); once()
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(3):
    once()
  torch.cuda.synchronize()
  dt = (time.perf_counter() - t0) / 3
  n = sum(n for _, n in assign)
  print(f"{assign}: {dt * 1e3:8.1f} ms per round, {n / dt:6.2f} images/s (UNet loop only)")


run([(0, 4)])
run([(0, 2), (1, 2)])
run([(0, 4)])
run([(0, 8)])
run([(0, 4), (1, 4)])
run([(0, 2)])
