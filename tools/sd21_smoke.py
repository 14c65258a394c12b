import sys, time, torch
sys.path.insert(0, "/root/repo")
sys.argv = ["x"]
import bench
import gill_amd
gill_amd.configure_hip_runtime()
from gill_amd import synth
from gill_amd.sd import GillSDPipeline
dev = torch.device("cuda:0")
cfg = synth.UNetConfig.sd21_768()
sd = bench.gpu_state_dict(lambda c, meta: bench.shapes_of("unet_state_dict", c), cfg, dev, 1)
uncond = synth.uncond_context(cfg.ctx_len, cfg.cross_attention_dim, 0)
pipe = GillSDPipeline(sd, cfg, uncond, dev, max_batch=8)
cond = torch.randn(4, 77, 1024, device=dev).bfloat16()
lat = synth.initial_latents(4, 4, 96, seed=1).to(dev)
out = pipe(prompt_embeds=cond, latents=lat, guidance_scale=7.5, num_inference_steps=4).images
torch.cuda.synchronize(); t0 = time.time()
out = pipe(prompt_embeds=cond, latents=lat, guidance_scale=7.5, num_inference_steps=10).images
torch.cuda.synchronize(); dt = time.time() - t0
print("SD-2.1-768 geometry, 4 prompts, 10 steps (11 UNet calls of batch 8): %.1f ms, finite=%s, rms=%.3f" % (dt * 1e3, bool(torch.isfinite(out).all()), out.float().pow(2).mean().sqrt().item()))
