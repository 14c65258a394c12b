#!/usr/bin/env python
"""Benchmark of the GILL image-generation hot path on MI355X (contract: see the task brief / DESIGN.md section 6).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`, one rank per GPU)

A "step" = one pass of the hot path over one batch of synthetic prompts: token ids -> frozen OPT-6.7B forward ->
8 [IMG] hidden states -> GILLMapper -> (B,77,768) -> SD-1.5 UNet CFG/PLMS loop (50 steps = 51 UNet calls of batch 2B)
-> final latents, all-gathered over ranks.  Workload at every N: BASELINE.json configs[1] per GPU (4 prompts/GPU,
weak scaling); inputs (ids, weights, initial latents) are resident in HBM before the timed region.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

UNET_TFLOP_PER_SAMPLE_FORWARD = 0.8032   # 401.6 GMAC, SD-1.5, 64x64 latents
PEAK_BF16_TFLOPS = 2500.0                # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)


def gpu_state_dict(builder, cfg, dev, seed):
  """Shapes/names from gill_amd.synth, values drawn on the GPU (fan-in scaled like synth; perf is value-independent
  but random non-zero operands keep the DVFS behaviour honest)."""
  import numpy as np
  proto = builder(cfg, meta=True)
  g = torch.Generator(device=dev).manual_seed(seed)
  sd = {}
  for k, shape in proto.items():
    if k.endswith(".weight") and len(shape) >= 2:
      std = 1.0 / float(np.sqrt(np.prod(shape[1:])))
      if "embed_tokens" in k:
        std = 0.5
      elif "embed_positions" in k:
        std = 0.25
      t = torch.randn(shape, device=dev, dtype=torch.float32, generator=g).mul_(std).to(torch.bfloat16)
    elif k.endswith(".weight"):   # norm scales
      t = (1.0 + 0.1 * torch.randn(shape, device=dev, generator=g)).to(torch.bfloat16)
    elif k == "query_embs":
      t = torch.randn(shape, device=dev, generator=g).to(torch.bfloat16)
    else:
      t = (0.02 * torch.randn(shape, device=dev, generator=g)).to(torch.bfloat16)
    sd[k] = t
  return sd


def shapes_of(builder_name, cfg):
  """state-dict inventory (name -> shape) without materialising values: run the synth builder with a stub generator."""
  from gill_amd import synth
  real = synth.normal
  shapes = {}

  class _Fake:
    def __init__(self, shape):
      self.shape = tuple(shape)

  def fake_normal(name, shape, seed, std=1.0, mean=0.0):
    shapes[name] = tuple(shape)
    return _Fake(shape)
  synth.normal = fake_normal
  try:
    getattr(synth, builder_name)(cfg, 0)
  finally:
    synth.normal = real
  return shapes


def build_model(dev, opt_cfg, unet_cfg, max_prompts):
  from types import SimpleNamespace
  from gill_amd import synth
  from gill_amd.models import GILL
  from gill_amd.sd import GillSDPipeline
  tok = synth.HashTokenizer()
  opt_sd = gpu_state_dict(lambda c, meta: shapes_of("opt_state_dict", c), opt_cfg, dev, 0)
  unet_sd = gpu_state_dict(lambda c, meta: shapes_of("unet_state_dict", c), unet_cfg, dev, 1)
  uncond = synth.uncond_context(unet_cfg.ctx_len, unet_cfg.cross_attention_dim, 0)
  vae_cfg = synth.VAEConfig.sd15()
  vae_sd = gpu_state_dict(lambda c, meta: shapes_of("vae_decoder_state_dict", c), vae_cfg, dev, 3)
  pipe = GillSDPipeline(unet_sd, unet_cfg, uncond, dev, max_batch=2 * min(8, max_prompts), vae_state=vae_sd, vae_cfg=vae_cfg)
  del unet_sd, vae_sd
  name = "facebook/opt-6.7b" if opt_cfg.hidden_size == 4096 else "facebook/opt-125m"
  args = SimpleNamespace(freeze_lm=True, freeze_vm=True, opt_version=name, visual_encoder="openai/clip-vit-large-patch14",
                         n_visual_tokens=4, ret_emb_dim=256, gen_emb_dim=768, text_emb_layers=[-1], text_fc_mode="gill_mapper",
                         ret_text_fc_mode="linear", num_tokens=8, num_clip_tokens=77, retrieval_token_idx=synth.IMG_TOKEN_IDS,
                         gen_token_idx=synth.IMG_TOKEN_IDS, opt_state_dict=opt_sd)
  g = GILL(tok, args, load_sd=True, sd_pipe=pipe)
  msd = gpu_state_dict(lambda c, meta: shapes_of("mapper_state_dict", c), synth.MapperConfig(in_dim=opt_cfg.hidden_size), dev, 2)
  g.model.gen_text_hidden_fcs[0].load_state_dict({k: v.float().cpu() for k, v in msd.items()}, strict=True)
  g = g.eval().bfloat16().cuda(dev)
  return g


def pmc_traffic_gb(prompts_per_step):
  """HBM bytes of one step from the committed PMC passes (profiles/r01_pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE /
  WRITE_SIZE over this very command, corrected as MI355X_MICROARCH.md prescribes), scaled to this run's prompts per step.
  bench.py cannot run a profiler around itself; None when the file is absent."""
  path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_pmc_traffic.json")
  if not os.path.exists(path):
    return None
  with open(path) as f:
    t = json.load(f)
  kb = t["FETCH_SIZE_kb_hot_path"] * t["gfx950_fetch_correction"] + t["WRITE_SIZE_kb_hot_path"]
  return kb * 1024.0 / 1e9 * (prompts_per_step / 4.0)


def kernel_rooflines(dev):
  """The three dominant device kernels of the UNet loop, each timed alone with HIP events on its level-0 shape of the CFG
  batch 8 through the operator C ABI (gill_op_*): algorithmic FLOPs / average launch time."""
  os.environ["GILL_OP_REPEAT"] = "20"      # read once by libgill_amd at the first gill_op_* call
  from gill_amd import ops
  out = []

  def timed(fn, flops, name):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    out.append({"kernel": name, "avg_launch_us": us, "achieved": flops / us / 1e6, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                "frac": flops / us / 1e6 / PEAK_BF16_TFLOPS, "bound": "mfma"})

  x = torch.randn(8, 64, 64, 320, device=dev).bfloat16()
  w = torch.randn(320, 320, 3, 3, device=dev) * 0.02
  timed(lambda: ops.conv3x3(x, w), 2.0 * 8 * 4096 * 320 * 2880, "gemm_kernel<160,1,0,2>: 3x3 conv 320->320 @ 64x64 x 8 (implicit GEMM)")
  q = torch.randn(8, 4096, 320, device=dev).bfloat16()
  timed(lambda: ops.attention(q, q, q, 8), 4.0 * 8 * 8 * 4096 * 4096 * 40, "attention_kernel<48>: self-attention N=4096, 8 heads x d=40, x 8 (incl. head re-layout wrappers)")
  a = torch.randn(32768, 1280, device=dev).bfloat16()
  wl = (torch.randn(320, 1280, device=dev) * 0.03).bfloat16()
  timed(lambda: ops.gemm(a, wl), 2.0 * 32768 * 320 * 1280, "gemm_kernel<160,0,0,2>: GEMM 32768 x 320 x 1280 (FF out projection)")
  return out


def cpu_baseline(n_infer_steps):
  """The CPU oracle timed on this host's cores on a bounded sample of the same workload: ONE full-size SD-1.5 UNet
  forward of the CFG pair (batch 2) + the GILLMapper, extrapolated to images/s (OPT-6.7b fp32 would need 27 GB and
  minutes; it is 0.65 % of an image's FLOPs and is left out of the sample, which flatters the CPU slightly)."""
  from gill_amd import synth
  from oracle import mapper_ref, unet_ref
  cores = synth.host_cores()
  torch.set_num_threads(cores)
  cfg = synth.UNetConfig.sd15()
  sd = synth.unet_state_dict(cfg, seed=0)
  x = synth.initial_latents(2, 4, 64)
  ctx = synth.normal("cpu_ctx", (2, 77, 768), 0)
  t0 = time.time()
  unet_ref.unet_forward(sd, x, torch.tensor([961.0, 961.0]), ctx)
  t_unet = time.time() - t0
  msd = synth.mapper_state_dict(synth.MapperConfig(in_dim=4096), seed=0)
  t0 = time.time()
  mapper_ref.mapper_forward(msd, synth.normal("cpu_x", (1, 8, 4096), 0), synth.normal("cpu_e", (1, 8, 4096), 0))
  t_map = time.time() - t0
  per_image = (n_infer_steps + 1) * t_unet + t_map
  return {"value": 1.0 / per_image, "unit": "images/s", "cores": cores, "kind": "port",
          "sample": f"1 SD-1.5 UNet forward of the CFG pair (batch 2, 1.61 TFLOP): {t_unet:.2f} s; GILLMapper B=1: {t_map * 1e3:.0f} ms; "
                    f"extrapolated x{n_infer_steps + 1} UNet calls per image (OPT forward excluded from the sample)"}


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--gpus", type=int, default=1)
  ap.add_argument("--steps", type=int, default=2)
  ap.add_argument("--warmup", type=int, default=1)
  ap.add_argument("--prompts-per-gpu", type=int, default=4)
  ap.add_argument("--infer-steps", type=int, default=50)
  ap.add_argument("--prompt-len", type=int, default=24)
  ap.add_argument("--small", action="store_true", help="opt-125m shapes (debug only; not the benchmark config)")
  ap.add_argument("--no-cpu-baseline", action="store_true")
  a = ap.parse_args()

  world = int(os.environ.get("WORLD_SIZE", "1"))
  rank = int(os.environ.get("RANK", "0"))
  local = int(os.environ.get("LOCAL_RANK", "0"))
  if world > 1:
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
  assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}"
  dev = torch.device("cuda", local)
  torch.cuda.set_device(dev)

  from gill_amd import synth
  opt_cfg = synth.OptConfig.opt_125m() if a.small else synth.OptConfig.opt_6_7b()
  unet_cfg = synth.UNetConfig.sd15()
  P = a.prompts_per_gpu
  g = build_model(dev, opt_cfg, unet_cfg, P)
  ids = synth.synthetic_prompt_ids(P * world, a.prompt_len, seed=0)[:, :a.prompt_len]   # [IMG] ids are appended by generate_images
  lat0 = synth.initial_latents(P * world, 4, unet_cfg.sample_size, seed=1337).to(dev)

  # HIP events around the UNet loop (same stream the kernels are launched on: torch's current stream)
  ev = {"t": []}
  orig_call = g.sd_pipe.__class__.__call__

  def timed_call(self, *args, **kw):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    out = orig_call(self, *args, **kw)
    e1.record()
    ev["t"].append((e0, e1))
    return out
  g.sd_pipe.__class__.__call__ = timed_call

  vae_ev = []
  orig_dec = g.sd_pipe.__class__.decode_latents

  def timed_dec(self, *args, **kw):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    o = orig_dec(self, *args, **kw)
    e1.record()
    vae_ev.append((e0, e1))
    return o
  g.sd_pipe.__class__.decode_latents = timed_dec

  def step():   # prompts -> OPT -> mapper -> 51 UNet calls -> latents (all-gathered) -> VAE decode -> uint8 512x512 images
    return g.generate_images(ids, num_inference_steps=a.infer_steps, guidance_scale=7.5, latents=lat0, decode=True)

  for _ in range(a.warmup):
    step()
  ev["t"].clear()
  vae_ev.clear()
  if world > 1:
    dist.barrier()
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(a.steps):
    out, images = step()
  if world > 1:
    dist.barrier()
  torch.cuda.synchronize()
  dt = time.perf_counter() - t0
  if world > 1:
    tt = torch.tensor([dt], device=dev, dtype=torch.float64)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt.item())
  assert out.shape == (P * world, 4, unet_cfg.sample_size, unet_cfg.sample_size) and bool(torch.isfinite(out).all())
  assert images.shape == (P, 512, 512, 3) and images.dtype == torch.uint8

  if rank == 0:
    images = P * world * a.steps
    value = images / dt
    unet_ms = sum(e0.elapsed_time(e1) for e0, e1 in ev["t"]) / max(1, len(ev["t"]))   # per sd_pipe call (P prompts)
    vae_ms = sum(e0.elapsed_time(e1) for e0, e1 in vae_ev) / max(1, len(vae_ev))
    flop_per_call = UNET_TFLOP_PER_SAMPLE_FORWARD * 2 * (a.infer_steps + 1) * P
    achieved = flop_per_call / (unet_ms * 1e-3)
    rec = {
      "metric": "512x512 images/sec/node, OPT-6.7B+SD1.5 50-step", "value": value, "unit": "images/s",
      "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
      "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
      "config": {"workload": f"{'opt-125m' if a.small else 'opt-6.7b'} + GILLMapper + SD-1.5 UNet (random-init weights of the "
                             f"exact shapes), {P} prompts/GPU x {world} GPU, prompt {a.prompt_len}+8 [IMG] tokens, "
                             f"{a.infer_steps} PLMS steps ({a.infer_steps + 1} UNet calls, CFG 7.5, batch {2 * P}), final latents "
                             f"all-gathered, then VAE decode of the local shard to uint8 512x512 ({vae_ms:.1f} ms per {P} images)", "parallelism": f"dp{world}"},
      "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                   "frac": achieved / PEAK_BF16_TFLOPS, "traffic": pmc_traffic_gb(P), "traffic_unit": "GB per step (PMC, profiles/r01_pmc_traffic.json)",
                   "kernel": "SD-1.5 UNet denoise loop (gill_sd_denoise: MFMA GEMM/implicit-conv + flash attention kernels)",
                   "algorithmic_tflop_per_launch": flop_per_call, "avg_launch_ms": unet_ms},
    }
    if world == 1 and not a.small:
      rec["roofline_kernels"] = kernel_rooflines(dev)
    if not a.no_cpu_baseline and world == 1:
      rec["cpu_baseline"] = cpu_baseline(a.infer_steps)
    else:
      rec["cpu_baseline"] = None
    print(json.dumps(rec), flush=True)
  if world > 1:
    dist.destroy_process_group()


if __name__ == "__main__":
  main()
