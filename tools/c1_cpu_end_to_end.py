#!/usr/bin/env python
"""BASELINE.json configs[0] (C1) end to end on the host cores: opt-125m-shape OPT -> 8 [IMG] hidden states -> GILLMapper -> SD-1.5 UNet
CFG / PLMS loop (10 steps = 11 UNet calls of the CFG pair) -> VAE decode to a 512x512 uint8 image, 1 prompt, fp32, through the CPU
oracle (oracle/: the reference's algorithm restated in plain PyTorch-CPU ops) — BASELINE.md section 3.2's "C1 — full end-to-end
timing", quoted in BASELINE.md section 4 beside the extrapolated `cpu_baseline` of bench.py.  Weights: gill_amd.synth (random-init, exact
shapes).  No GPU involved.   python tools/c1_cpu_end_to_end.py [--steps 10]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gill_amd import synth  # noqa: E402
from oracle import pipeline_ref, vae_ref  # noqa: E402


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--steps", type=int, default=10)
  ap.add_argument("--prompt-len", type=int, default=24)
  a = ap.parse_args()
  cores = synth.host_cores()
  torch.set_num_threads(cores)
  t_all = time.time()
  opt_cfg = synth.OptConfig.opt_125m()
  unet_cfg = synth.UNetConfig.sd15()
  opt_sd = synth.opt_state_dict(opt_cfg, seed=0)
  msd = synth.mapper_state_dict(synth.MapperConfig(in_dim=opt_cfg.hidden_size, out_dim=unet_cfg.cross_attention_dim), seed=0)
  usd = synth.unet_state_dict(unet_cfg, seed=1)
  vsd = synth.vae_decoder_state_dict(synth.VAEConfig.sd15(), seed=3)
  uncond = synth.uncond_context(unet_cfg.ctx_len, unet_cfg.cross_attention_dim, seed=0)
  t_build = time.time() - t_all
  ids = synth.synthetic_prompt_ids(1, a.prompt_len, seed=0)[:, :a.prompt_len]
  img = torch.tensor(synth.IMG_TOKEN_IDS, dtype=torch.int64)
  full = torch.cat([ids[0], img]).unsqueeze(0)
  last_idx = torch.tensor([full.shape[1] - 1])
  with torch.no_grad():
    t0 = time.time()
    cond = pipeline_ref.sd_embedding(opt_sd, msd, opt_cfg.num_layers, opt_cfg.num_heads, full, last_idx)
    t_stage12 = time.time() - t0
    lat0 = synth.initial_latents(1, 4, unet_cfg.sample_size, seed=1337)
    t0 = time.time()
    lat = pipeline_ref.denoise(usd, cond, uncond, lat0, a.steps, 7.5, unet_cfg.block_out_channels, unet_cfg.num_heads, unet_cfg.norm_num_groups)
    t_loop = time.time() - t0
    t0 = time.time()
    image = vae_ref.to_uint8(vae_ref.vae_decode(vsd, lat))
    t_vae = time.time() - t0
  total = t_stage12 + t_loop + t_vae
  assert tuple(image.shape[-3:]) in ((512, 512, 3), (3, 512, 512)) and bool(torch.isfinite(lat).all())
  print(json.dumps({"config": "C1: opt-125m shapes + GILLMapper + SD-1.5 UNet + VAE decoder, 1 prompt, %d PLMS steps (%d UNet calls of the CFG pair), fp32, CPU oracle" % (a.steps, a.steps + 1),
                    "cores": cores, "images_per_s": 1.0 / total, "s_per_image": total, "opt_and_mapper_s": t_stage12, "unet_loop_s": t_loop,
                    "s_per_unet_call": t_loop / (a.steps + 1), "vae_decode_s": t_vae, "weight_build_s": t_build}))


if __name__ == "__main__":
  main()
