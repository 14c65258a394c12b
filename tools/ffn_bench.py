#!/usr/bin/env python3
"""Operator-level timing of the fused feed-forward block (csrc/ffn.hip) at the level-0 size of the CFG batch 8 (M = 32768), against the
GEGLU + two-source GEMM pair it replaces is in profiles/r03_forward_timeline_c.txt (97 + 50 us).   python tools/ffn_bench.py [M]"""
import os
import sys
os.environ.setdefault("GILL_OP_REPEAT", "20")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gill_amd import ops

M = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
dev = torch.device("cuda:0")
C, H = 320, 1280
g = torch.Generator().manual_seed(1)
r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc)
t, resid = r(M, C).bfloat16().to(dev), r(M, C).bfloat16().to(dev)
args = [t, (1 + 0.1 * r(C)).to(dev), (0.1 * r(C)).to(dev), r(2 * H, C, sc=0.06).bfloat16().to(dev), (0.1 * r(2 * H)).to(dev),
        r(C, H, sc=0.03).bfloat16().to(dev), (0.1 * r(C)).to(dev), r(C, C, sc=0.05).bfloat16().to(dev), (0.1 * r(C)).to(dev), resid]
ops.ffn_fused(*args, rows_per_batch=4096); torch.cuda.synchronize()
best = 1e9
for _ in range(3):
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record(); ops.ffn_fused(*args, rows_per_batch=4096); e1.record(); torch.cuda.synchronize()
  best = min(best, e0.elapsed_time(e1) * 1e3 / int(os.environ["GILL_OP_REPEAT"]))
flop = 2.0 * M * (2 * H * C + C * H + C * C)
print(f"ffn_fused M={M}: {best:.1f} us per launch (incl. 1/20 of the weight preparation), {flop / best / 1e6:.0f} TFLOP/s")
