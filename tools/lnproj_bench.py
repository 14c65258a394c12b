#!/usr/bin/env python3
"""Operator-level timing of the fused projection pairs around norm1 / norm2 (csrc/lnproj.hip) at the level-0 size of the CFG batch 8
(M = 32768).  The GEMM pairs they replace are in profiles/r03_forward_timeline_d.txt.   python tools/lnproj_bench.py [B] [HW]"""
import os
import sys
os.environ.setdefault("GILL_OP_REPEAT", "50")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gill_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
HW = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
M, C = B * HW, 320
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc)
rep = int(os.environ["GILL_OP_REPEAT"])
for mode in (0, 1):
  nseg = 3 if mode == 0 else 1
  x, t = r(M, C).bfloat16().to(dev), r(M, C).bfloat16().to(dev)
  args = [mode, x, t, r(C, C, sc=0.08).bfloat16().to(dev), (0.1 * r(C)).to(dev), (1 + 0.1 * r(C)).to(dev), (0.1 * r(C)).to(dev),
          r(nseg * C, C, sc=0.06).bfloat16().to(dev), B, HW]
  ops.lnproj(*args); torch.cuda.synchronize()
  best = 1e9
  for _ in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ops.lnproj(*args); e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) * 1e3 / rep)
  k1 = 320 if mode == 0 else 384
  flop = 2.0 * M * (C * k1 + nseg * 384 * C)
  print(f"lnproj mode {mode} M={M}: {best:.1f} us per launch (incl. 1/{rep} of the operand preparation), {flop / best / 1e6:.0f} TFLOP/s")
