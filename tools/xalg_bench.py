#!/usr/bin/env python
"""Op-level timing of the two cross-attention GEMMs (gill_op_cross_attention_folded) at the UNet's level 1 / 2 / mid shapes of the
8-sample batch: run under `rocprofv3 --kernel-trace --stats` with GILL_OP_REPEAT=20 and read the gemm_kernel<..., 5, ...> rows."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gill_amd import ops
g = torch.Generator().manual_seed(0)
r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).cuda()
b = lambda *s, sc=1.0: r(*s, sc=sc).bfloat16()
for B, HW, C in ((8, 1024, 640), (8, 256, 1280), (8, 64, 1280)):
  out, P = ops.cross_attention_folded(b(B * HW, C), 1 + 0.1 * r(C), 0.1 * r(C), b(C, C, sc=0.05), b(C, 768, sc=0.05), b(C, 768, sc=0.05),
                                      b(C, C, sc=0.05), r(C), b(B, 77, 768), 8, B, HW)
  torch.cuda.synchronize()
  print(B, HW, C, float(out.float().abs().mean()))
