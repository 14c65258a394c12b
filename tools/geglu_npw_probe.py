#!/usr/bin/env python
"""Probe: N tiles per workgroup of the GEGLU GEMM (builds with -DGILL_NPW_FORCE=n) on the UNet's level-1 / level-2 shapes.  Per-call time by repeat differencing."""
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
for (M, inner, K) in [(2048, 5120, 1280), (8192, 2560, 640), (512, 5120, 1280)]:
  a = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(2 * inner, K, device=dev) * 0.03).bfloat16(); b = torch.randn(2 * inner, device=dev)
  t = per_call(lambda: ops.geglu(a, w, b))
  print(f"GEGLU {M} x {2 * inner} x {K}: {t:7.1f} us")
