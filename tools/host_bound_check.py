"""Is the denoise loop host-bound?  Compares the host time to ENQUEUE gill_sd_denoise with the time to completion."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import gill_amd
gill_amd.configure_hip_runtime()
from gill_amd import synth
from gill_amd.sd import GillSDPipeline

dev = torch.device("cuda:0")
cfg = synth.UNetConfig.sd15()
sd = bench.gpu_state_dict(lambda c, meta: bench.shapes_of("unet_state_dict", c), cfg, dev, 1)
pipe = GillSDPipeline(sd, cfg, synth.uncond_context(), dev, max_batch=16)
for B in (1, 4, 8):
  cond = torch.randn(B, 77, 768, device=dev).bfloat16()
  lat = synth.initial_latents(B).to(dev)
  pipe(prompt_embeds=cond, latents=lat, num_inference_steps=10)
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  pipe(prompt_embeds=cond, latents=lat, num_inference_steps=50)
  t1 = time.perf_counter()
  torch.cuda.synchronize()
  t2 = time.perf_counter()
  print(f"B={B}: enqueue {1e3*(t1-t0):.1f} ms, complete {1e3*(t2-t0):.1f} ms  -> {B/(t2-t0):.2f} img/s")
