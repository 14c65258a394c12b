#!/usr/bin/env python
"""Probe: short-K, many-tile plain GEMMs (the GEGLU / QKV shapes of UNet levels 1-2, without their epilogues) on the general tiles vs on the ping-pong
256 / 128 x 160 tiles, one tile per workgroup (a build with -DGILL_PP_KMIN=320 admits them).  Per-call time by repeat differencing."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gill_amd
gill_amd.configure_hip_runtime()
from gill_amd import ops
dev = torch.device("cuda:0")
def per_call(fn, r1=4, r2=24):
  ms = {}
  for r in (r1, r2, r1, r2):
    os.environ["GILL_OP_REPEAT"] = str(r)
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    ms[r] = e0.elapsed_time(e1)
  os.environ["GILL_OP_REPEAT"] = "1"
  return (ms[r2] - ms[r1]) * 1e3 / (r2 - r1)
for (M, N, K) in [(8192, 5120, 640), (2048, 10240, 1280), (8192, 1920, 640), (2048, 3840, 1280), (8192, 640, 2560), (8192, 640, 640), (2048, 1280, 1280), (32768, 320, 320), (32768, 320, 1280), (512, 1280, 1280)]:
  a = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(N, K, device=dev) * 0.03).bfloat16(); b = torch.randn(N, device=dev)
  r = torch.randn(M, N, device=dev).bfloat16()
  t = per_call(lambda: ops.gemm(a, w, b, r))
  print(f"GEMM {M} x {N} x {K}: {t:7.1f} us  {2.0 * M * N * K / t / 1e6:6.0f} TFLOP/s")
