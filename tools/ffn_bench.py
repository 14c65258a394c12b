#!/usr/bin/env python3
"""Operator-level timing of the fused feed-forward block (csrc/ffn.hip) at the level-0 size of the CFG batch 8 (M = 32768): the plain form
and the form with attn2.to_out + residual in front (PRE: what the UNet engine runs).  The kernels they replace are in
profiles/r03_forward_timeline_c.txt (GEGLU 97 + ffo GEMM 50 us; attn2.to_out GEMM 23-26 us).   python tools/ffn_bench.py [M]"""
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
pre = dict(o2=r(M, C).bfloat16().to(dev), wo=r(C, C, sc=0.06).bfloat16().to(dev), bo2=(0.1 * r(C)).to(dev))
rep = int(os.environ["GILL_OP_REPEAT"])
for name, kw in (("plain", {}), ("with attn2.to_out", pre)):
  ops.ffn_fused(*args, rows_per_batch=4096, **kw); torch.cuda.synchronize()
  best = 1e9
  for _ in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ops.ffn_fused(*args, rows_per_batch=4096, **kw); e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) * 1e3 / rep)
  flop = 2.0 * M * (2 * H * C + C * H + C * C + (C * 384 if kw else 0))
  print(f"ffn_fused {name} M={M}: {best:.1f} us per launch (incl. 1/{rep} of the operand preparation), {flop / best / 1e6:.0f} TFLOP/s")
