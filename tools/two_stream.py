"""Does running the CFG batch as TWO independent half batches on two streams (two UNet handles, kernels of the two chains free to
overlap each other's tails and ramps) beat one handle on the full batch?   python tools/two_stream.py [prompts=4] [steps=50]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gill_amd
gill_amd.configure_hip_runtime()
from gill_amd import synth
from gill_amd.sd import GillSDPipeline

P = int(sys.argv[1]) if len(sys.argv) > 1 else 4
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
dev = torch.device("cuda:0")
cfg = synth.UNetConfig.sd15()
sd = {k: v.bfloat16() for k, v in synth.unet_state_dict(cfg, seed=41).items()}
uncond = synth.uncond_context(cfg.ctx_len, cfg.cross_attention_dim, seed=41)
cond = synth.normal("ts_cond", (P, cfg.ctx_len, cfg.cross_attention_dim), 42).bfloat16().to(dev)
lat0 = synth.initial_latents(P, 4, cfg.sample_size, seed=7).to(dev)
full = GillSDPipeline(sd, cfg, uncond, dev, max_batch=2 * P)
halves = [GillSDPipeline(sd, cfg, uncond, dev, max_batch=P) for _ in range(2)]
streams = [torch.cuda.Stream(dev) for _ in range(2)]
kw = dict(guidance_scale=7.5, num_inference_steps=steps, output_type="latent")

def run_full():
  return full(prompt_embeds=cond, latents=lat0, **kw).images

def run_halves():
  outs = []
  cur = torch.cuda.current_stream()
  for i, (p, s) in enumerate(zip(halves, streams)):
    s.wait_stream(cur)
    with torch.cuda.stream(s):
      outs.append(p(prompt_embeds=cond[i * P // 2:(i + 1) * P // 2], latents=lat0[i * P // 2:(i + 1) * P // 2], **kw).images)
  for s in streams:
    cur.wait_stream(s)
  return torch.cat(outs, 0)

for name, fn in (("one handle, batch %d" % P, run_full), ("two handles x batch %d on two streams" % (P // 2), run_halves), ("one handle again", run_full)):
  ref = fn(); torch.cuda.synchronize()
  t0 = time.time()
  for _ in range(3):
    out = fn()
  torch.cuda.synchronize()
  dt = (time.time() - t0) / 3
  print(f"{name}: {dt * 1e3:.1f} ms per {P}-prompt loop, finite={bool(torch.isfinite(out).all())}")
a = run_full().float(); b = run_halves().float(); torch.cuda.synchronize()
print("rel-L2 halves vs full:", ((a - b).norm() / a.norm()).item())
