#!/usr/bin/env python3
"""Split-K sweep of a plain GEMM with residual at the engine's long-K shapes (feed-forward output GEMMs of levels 1-3): time per launch
(incl. the reducer) for splitk = 1, 2, 3, 4 and the heuristic's own choice (0).   python tools/gemm_sk_sweep.py"""
import os
import sys
os.environ.setdefault("GILL_OP_REPEAT", "50")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gill_amd import ops

dev = torch.device("cuda:0")
rep = int(os.environ["GILL_OP_REPEAT"])
g = torch.Generator().manual_seed(1)
for (M, N, K) in ((8192, 640, 3200), (2048, 1280, 6400), (512, 1280, 6400), (2048, 1280, 1280), (8192, 640, 640)):
  a = (torch.randn(M, K, generator=g)).bfloat16().to(dev)
  w = (torch.randn(N, K, generator=g) * 0.05).bfloat16().to(dev)
  b = torch.randn(N, generator=g).to(dev)
  r = torch.randn(M, N, generator=g).bfloat16().to(dev)
  row = []
  for sk in (0, 1, 2, 3, 4, 6):
    ops.gemm(a, w, b, resid=r, splitk=sk); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
      e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      e0.record(); ops.gemm(a, w, b, resid=r, splitk=sk); e1.record(); torch.cuda.synchronize()
      best = min(best, e0.elapsed_time(e1) * 1e3 / rep)
    row.append(f"sk{sk}: {best:6.1f}")
  print(f"{M:6d} x {N:5d} x {K:5d}  " + "  ".join(row) + f"   us  ({2.0 * M * N * K / 1e9:.1f} GFLOP)")
