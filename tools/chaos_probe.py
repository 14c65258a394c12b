import os, subprocess, sys, torch
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = ("import hashlib, os, sys, torch; sys.path.insert(0, %r)\n"
        "from gill_amd import synth\n"
        "from gill_amd.sd import GillSDPipeline\n"
        "cfg = synth.UNetConfig.sd15()\n"
        "sd = {k: v.bfloat16() for k, v in synth.unet_state_dict(cfg, seed=41).items()}\n"
        "uncond = synth.uncond_context(cfg.ctx_len, cfg.cross_attention_dim, seed=41)\n"
        "pipe = GillSDPipeline(sd, cfg, uncond, 'cuda:0', max_batch=4)\n"
        "cond = synth.normal('pp_cond', (2, 77, 768), 42).bfloat16()\n"
        "lat0 = synth.initial_latents(2, 4, 64, seed=11)\n"
        "lat = pipe(prompt_embeds=cond, latents=lat0, guidance_scale=7.5, num_inference_steps=int(os.environ.get('NSTEPS','3')), output_type='latent').images\n"
        "torch.save(lat.float().cpu(), os.environ['GILL_TEST_OUT'])\n") % root
def run(tag, **env):
  out = f"/tmp/chaos_{tag}.pt"
  e = dict(os.environ, GILL_CONV_KORDER="0", GILL_TEST_OUT=out, **env)
  r = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, timeout=900)
  assert r.returncode == 0, r.stderr[-1500:]
  return torch.load(out)
base = run("base", GILL_GEMM_PP="0", GILL_GEMM_PP128="0")
for tag, env in (("pp256", dict(GILL_GEMM_PP="1", GILL_GEMM_PP128="0")), ("pp128", dict(GILL_GEMM_PP="1", GILL_GEMM_PP128="1")),
                 ("minsteps48", dict(GILL_GEMM_PP="0", GILL_GEMM_PP128="0", GILL_GEMM_MINSTEPS="48")),
                 ("minsteps12", dict(GILL_GEMM_PP="0", GILL_GEMM_PP128="0", GILL_GEMM_MINSTEPS="12"))):
  x = run(tag, **env)
  print(tag, "rel-L2 vs base:", ((x - base).norm() / base.norm()).item())
