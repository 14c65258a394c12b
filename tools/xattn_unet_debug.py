import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gill_amd import synth
from gill_amd.sd import GillSDPipeline
cfg = synth.UNetConfig.sd15()
sd = {k: v.bfloat16().float() for k, v in synth.unet_state_dict(cfg, seed=0).items()}
uncond = synth.uncond_context(seed=0)
pipe = GillSDPipeline(sd, cfg, uncond, "cuda:0", max_batch=2)
x = synth.initial_latents(2, 4, 64, seed=1337)
ctx = torch.cat([uncond, synth.normal("sd15_ctx", (1, 77, 768), 2)], 0).bfloat16().float()
got = pipe.unet(x, torch.tensor([961.0, 961.0]), ctx)
torch.cuda.synchronize()
print("forward ok", bool(torch.isfinite(got).all()), float(got.abs().mean()))
