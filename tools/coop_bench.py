#!/usr/bin/env python
"""Operator-level timing of the round-6 in-kernel finishes (gemm.hip "COOP") against the launches they replace, on the UNet's shapes at the CFG
batch 8: per shape the per-call time of gill_op_conv3x3_gn with coop=0 (split-K: conv + reducer that normalises; splitk 1: conv + GroupNorm-apply)
and coop=1 (one launch).  Per-call time = (call at R2 repeats - call at R1 repeats) / (R2 - R1), HIP events (the wrapper's own work cancels).
  python tools/coop_bench.py            (GILL_XMAP=0 in the environment: the XCD range map instead of the block map)"""
import os
import sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gill_amd
gill_amd.configure_hip_runtime()
from gill_amd import ops   # noqa: E402

dev = torch.device("cuda:0")
R1, R2 = 4, 24


def per_call(fn):
  ms = {}
  for r in (R1, R2, R1, R2):
    os.environ["GILL_OP_REPEAT"] = str(r)
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    ms[r] = e0.elapsed_time(e1)
  os.environ["GILL_OP_REPEAT"] = "1"
  return (ms[R2] - ms[R1]) * 1e3 / (R2 - R1)


shapes = [   # B, H, W, Cin, Cout, splitk, note
  (8, 8, 8, 1280, 1280, 8, "level 3 conv (mid / down3 resnets)"),
  (8, 8, 8, 2560, 1280, 8, "level 3 up-block conv1 (concat input)"),
  (8, 16, 16, 1280, 1280, 2, "level 2 conv"),
  (8, 16, 16, 2560, 1280, 2, "level 2 up-block conv1 (concat input)"),
  (8, 16, 16, 640, 1280, 2, "level 2 down-block conv1"),
  (8, 32, 32, 640, 640, 1, "level 1 conv"),
  (8, 32, 32, 1280, 640, 1, "level 1 up-block conv1"),
  (8, 64, 64, 320, 320, 1, "level 0 conv"),
  (8, 64, 64, 640, 320, 1, "level 0 up-block conv1"),
]
print(f"GILL_XMAP={os.environ.get('GILL_XMAP', '(default: block map)')}")
for (B, H, W, Cin, Cout, sk, note) in shapes:
  x = torch.randn(B, H, W, Cin, device=dev).bfloat16()
  w = torch.randn(Cout, Cin, 3, 3, device=dev) * (9 * Cin) ** -0.5
  b = torch.randn(Cout, device=dev) * 0.1
  g = torch.ones(Cout, device=dev); be = torch.zeros(Cout, device=dev)
  rv = torch.randn(B, Cout, device=dev) * 0.3
  t = {}
  for coop in (False, True):
    t[coop] = per_call(lambda: ops.conv3x3_gn(x, w, b, g, be, 32, 1e-5, True, None, sk, False, coop=coop, rowvec=rv))
  fl = 2.0 * B * H * W * Cout * 9 * Cin
  print(f"  {note:40s} {B}x{H}x{W} {Cin:4d}->{Cout:4d} sk{sk}: launches {t[False]:7.1f} us   in-kernel {t[True]:7.1f} us   ({t[True] - t[False]:+6.1f} us; "
        f"{fl / t[True] / 1e6:5.0f} TFLOP/s)")
