#!/usr/bin/env python3
"""Round-5 probe: operator timings behind three decisions (one process, GILL_OP_REPEAT launches per timing, best of 3).

  1. Winograd F(2x2,3x3) go / no-go, performance half: the direct 3x3 convolution against an OPTIMISTIC stand-in for
     the Winograd GEMM work on the same layer — the 16 per-position [tiles x Cin].[Cin x Cout] GEMMs as ONE plain GEMM of 16 x tiles rows
     (same weights for every position, no input / output transform passes, no 4x transformed-input traffic).
  2. level-3 convolutions (8 x 8 maps, weight-bound): time vs split-K.
  3. OPT-6.7b weight-streaming GEMMs at M = 128 / 256 rows: time vs split-K.
  4. `optcold`: the same GEMMs on COLD weights — every launch streams another copy of the matrix, > 1 GB in rotation, so neither the 256 MB
     MALL nor an L2 serves a weight twice (mode 3 re-launches one 33-134 MB matrix 30 times: warm in the MALL, it overstates HBM streaming).
  5. `fp8lin`: an OPTIMISTIC stand-in for fp8 transformer linears: the bf16 GEMM of every level-1..3 linear at its own K and
     at K / 2 (an fp8 K walk has half the LDS-DMA pieces, LDS reads and MFMA issue slots of the bf16 one; prologue, epilogue and launch stay):
     sum of t(K) - t(K / 2) over a forward = the most fp8 linears could save, before any quantisation pass.
"""
import os
import sys
os.environ.setdefault("GILL_OP_REPEAT", "30")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gill_amd
gill_amd.configure_hip_runtime()
from gill_amd import ops

dev = torch.device("cuda:0")
rep = int(os.environ["GILL_OP_REPEAT"])


def timeit(fn):
  fn(); torch.cuda.synchronize()
  best = 1e9
  for _ in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) * 1e3 / rep)
  return best


def conv(B, H, W, C1, C2, Cout, sk=0):
  x1 = torch.randn(B, H, W, C1, device=dev).bfloat16()
  x2 = torch.randn(B, H, W, C2, device=dev).bfloat16() if C2 else None
  w = torch.randn(Cout, C1 + C2, 3, 3, device=dev) * 0.02
  return timeit(lambda: ops.conv3x3(x1, w, x2=x2, splitk=sk))


def gemm(M, N, K, sk=0, resid=False, out_f32=False):
  a = torch.randn(M, K, device=dev).bfloat16()
  w = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
  b = torch.randn(N, device=dev)
  return timeit(lambda: ops.gemm(a, w, b, splitk=sk, out_f32=out_f32))


what = sys.argv[1:] or ["wino", "l3", "opt"]
# ("winoop": the real Winograd operator against the direct convolution — profiles/r05_winograd_op_timing.log — went with csrc/wino.hip in round 6)
if "wino" in what:
  print("== Winograd stand-in: direct conv vs ONE plain GEMM of the 16 per-position GEMMs' rows (optimistic: no transforms)")
  for (B, H, W, C1, C2, Co) in ((8, 64, 64, 640, 320, 320), (8, 32, 32, 1280, 640, 640), (8, 32, 32, 640, 0, 640), (8, 16, 16, 1280, 0, 1280), (8, 16, 16, 1280, 1280, 1280)):
    tc = conv(B, H, W, C1, C2, Co)
    T = B * H * W // 4
    tg = gemm(16 * T, Co, C1 + C2)
    fl = 2.0 * B * H * W * Co * 9 * (C1 + C2)
    print(f"  conv {B}x{H}x{W} {C1 + C2:4d}->{Co:4d}: direct {tc:7.1f} us ({fl / tc / 1e6:5.0f} TFLOP/s)   GEMM {16 * T} x {Co} x {C1 + C2}: {tg:7.1f} us = {tg / tc:.2f} of direct")
if "l3" in what:
  print("== level-3 convolutions (8 samples x 8 x 8), time incl. reducer vs split-K (0 = heuristic)")
  for (C1, C2, Co) in ((1280, 0, 1280), (1280, 1280, 1280)):
    row = [f"sk{sk}: {conv(8, 8, 8, C1, C2, Co, sk):6.1f}" for sk in (0, 2, 4, 8, 16)]
    print(f"  {C1 + C2}->{Co}: " + "  ".join(row) + f"  us   (weights {Co * 9 * (C1 + C2) * 2 / 1e6:.1f} MB)")
if "opt" in what:
  print("== OPT-6.7b GEMMs (weight streaming), time incl. reducer vs split-K (0 = heuristic); GB/s = weight bytes / time")
  for M in (128, 256):
    for (N, K) in ((12288, 4096), (4096, 4096), (16384, 4096), (4096, 16384)):
      row = []
      for sk in (0, 1, 2, 3, 4, 6, 8, 16):
        t = gemm(M, N, K, sk, out_f32=(N == 4096))
        row.append(f"sk{sk}: {t:6.1f}")
      t0 = float(row[0].split(":")[1])
      print(f"  {M:4d} x {N:5d} x {K:5d}  " + "  ".join(row) + f"  us   heuristic = {N * K * 2 / t0 / 1e3:6.0f} GB/s")

if "optcold" in what:
  os.environ["GILL_OP_REPEAT"] = "1"
  print("== OPT-6.7b GEMMs on cold weights (copies in rotation, > 1 GB), incl. reducer; GB/s = weight bytes / time")
  for M in (128, 256, 32, 8):
    for (N, K) in ((12288, 4096), (4096, 4096), (16384, 4096), (4096, 16384)):
      ncopy = max(3, int(1.3e9 // (N * K * 2)) + 1)
      ws = [(torch.randn(N, K, device=dev) * 0.05).bfloat16() for _ in range(ncopy)]
      a = torch.randn(M, K, device=dev).bfloat16()
      b = torch.randn(N, device=dev)
      row = []
      for sk in (0, 1, 2, 4, 8):
        def run():
          for w in ws:
            ops.gemm(a, w, b, splitk=sk, out_f32=(N == 4096))
        run(); torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
          e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
          e0.record(); run(); e1.record(); torch.cuda.synchronize()
          best = min(best, e0.elapsed_time(e1) * 1e3 / ncopy)
        row.append((sk, best))
      del ws
      print(f"  {M:4d} x {N:5d} x {K:5d}  " + "  ".join(f"sk{sk}: {t:6.1f}" for sk, t in row) + f"  us   heuristic = {N * K * 2 / row[0][1] / 1e3:6.0f} GB/s")

if "fp8lin" in what:
  for B in (8, 32):
    print(f"== fp8-linear stand-in, UNet batch {B}: bf16 GEMM at K vs K / 2 (us), per block and per forward (5 level-1 + 5 level-2 + 1 mid block)")
    tot = 0.0
    for (name, HW, C, nblk) in (("level 1", 1024, 640, 5), ("level 2", 256, 1280, 5), ("mid", 64, 1280, 1)):
      M = B * HW
      rows = []
      for (what_, N, K, geglu) in (("qkv", 3 * C, C, False), ("geglu", 8 * C, C, True), ("ffo+proj_out", C, 5 * C, False), ("proj_in/to_out x2", C, C, False)):
        def t_at(k):
          a = torch.randn(M, k, device=dev).bfloat16()
          if geglu:
            w = (torch.randn(N, k, device=dev) * 0.05).bfloat16(); b = torch.randn(N, device=dev)
            return timeit(lambda: ops.geglu(a, w, b))
          w = (torch.randn(N, k, device=dev) * 0.05).bfloat16(); b = torch.randn(N, device=dev)
          return timeit(lambda: ops.gemm(a, w, b))
        tk, th = t_at(K), t_at(K // 2 // 64 * 64)
        mult = 2 if what_.startswith("proj_in") else 1
        rows.append((what_, tk, th, mult))
        tot += nblk * mult * (tk - th)
      print(f"  {name} (M = {M}, C = {C}): " + "   ".join(f"{w_} {tk:6.1f} -> {th:6.1f}" for w_, tk, th, _ in rows))
    print(f"  sum over a forward of t(K) - t(K/2): {tot:7.1f} us")
