#!/usr/bin/env python3
"""Fixed vs per-K-step cost of the implicit-GEMM 3x3 convolution: gill_op_conv3x3 at Cin = 64 .. 640 (9 K steps per 64 channels)
on a fixed output grid, timed with HIP events over GILL_OP_REPEAT launches; least-squares line through (K steps, us).
  python tools/conv_ksweep.py [B H W Cout]"""
import os
import sys
os.environ.setdefault("GILL_OP_REPEAT", "20")
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gill_amd import ops

B, H, W, Cout = (int(v) for v in sys.argv[1:5]) if len(sys.argv) >= 5 else (8, 64, 64, 320)
dev = torch.device("cuda:0")
rep = int(os.environ["GILL_OP_REPEAT"])
pts = []
for cin in (64, 128, 192, 256, 320, 448, 640):
  x = torch.randn(B, H, W, cin, device=dev).bfloat16()
  w = torch.randn(Cout, cin, 3, 3, device=dev) * 0.02
  ops.conv3x3(x, w, splitk=1); torch.cuda.synchronize()
  best = 1e9
  for _ in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ops.conv3x3(x, w, splitk=1); e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) * 1e3 / rep)
  steps = 9 * cin // 64
  fl = 2.0 * B * H * W * Cout * 9 * cin
  pts.append((steps, best))
  print(f"Cin {cin:4d}  K steps {steps:3d}  {best:7.1f} us  {fl / best / 1e6:7.1f} TFLOP/s")
a, b = np.polyfit([p[0] for p in pts], [p[1] for p in pts], 1)
fl_step = 2.0 * B * H * W * Cout * 64
print(f"fit: {b:.1f} us fixed + {a:.3f} us per K step  (main loop alone = {fl_step / a / 1e6:.0f} TFLOP/s)")
