#!/usr/bin/env python
"""Where a GEMM / convolution launch spends its time, by wall-clock stamps taken inside the workgroups (measurement-only build:
  patch -p0 < tools/patches/gemm_stamps.patch; tools/build_variant.sh stamps -DGILL_STAMPS; patch -R -p0 < tools/patches/gemm_stamps.patch;
  GILL_AMD_LIB=tools/_lib_stamps.so python tools/stamps.py).  Record: profiles/r06_launch_skeleton.md.
Stamps (thread 0 of every workgroup, 100 MHz wall clock): 0 tile entry (after the kernarg warm-up), 1 set-up done, 2 prologue stages issued,
3 first stage landed (ping-pong tiles), 4 main loop done, 5 epilogue loads landed, 6 GroupNorm partials published, 7 group complete,
8 last store issued, 9 stores acknowledged.  Printed: per stamp the mean / min / max over workgroups of (stamp - earliest entry), in us, next to
the launch's HIP-event time (back-to-back repeats of the same launch)."""
import ctypes
import os
import sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gill_amd
gill_amd.configure_hip_runtime()
from gill_amd import ops, _native   # noqa: E402

lib = ctypes.CDLL(_native.LIB_PATH)
dev = torch.device("cuda:0")
NAMES = ["entry", "set-up done", "prologue issued", "first stage landed", "main loop done", "epilogue loads landed", "partials published",
         "group complete", "last store issued", "stores acknowledged"]


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


def report(title, fn, nwg):
  us = per_call(fn)
  os.environ["GILL_OP_REPEAT"] = "3"
  zero = np.zeros(512 * 12, dtype=np.uint64)
  fn(); torch.cuda.synchronize()
  buf = np.zeros(512 * 12, dtype=np.uint64)
  assert lib.gill_debug_stamps(buf.ctypes.data_as(ctypes.POINTER(ctypes.c_ulonglong))) == 0
  os.environ["GILL_OP_REPEAT"] = "1"
  st = buf.reshape(512, 12)[:nwg].astype(np.int64)
  t0 = st[:, 0].min()
  print(f"\n== {title}: {us:.1f} us per launch (HIP events, back to back); {nwg} workgroups; entry skew {(st[:, 0].max() - t0) / 100:.2f} us")
  prev = None
  for k in range(10):
    col = st[:, k]
    if col.max() < t0:        # not written by this kernel (stale / zero)
      continue
    rel = (col - t0) / 100.0
    d = "" if prev is None else f"   (+{rel.mean() - prev:5.2f})"
    print(f"  {k} {NAMES[k]:24s} mean {rel.mean():7.2f}  min {rel.min():7.2f}  max {rel.max():7.2f}{d}")
    prev = rel.mean()
  print(f"  launch - last stamp = {us - (st[:, 9].max() - t0) / 100:.2f} us  (dispatch, wave launch, kernarg warm-up in front of stamp 0; drain and completion behind stamp 9)")


def conv_case(B, H, W, Cin, Cout, coop, rv=True, resid=False):
  x = torch.randn(B, H, W, Cin, device=dev).bfloat16()
  w = torch.randn(Cout, Cin, 3, 3, device=dev) * (9 * Cin) ** -0.5
  b = torch.randn(Cout, device=dev) * 0.1
  g = torch.ones(Cout, device=dev); be = torch.zeros(Cout, device=dev)
  rvec = torch.randn(B, Cout, device=dev) * 0.3 if rv else None
  rs = torch.randn(B, H, W, Cout, device=dev).bfloat16() if resid else None
  return lambda: ops.conv3x3_gn(x, w, b, g, be, 32, 1e-5, True, rs, 1, False, coop=coop, rowvec=rvec)


report("level-0 conv 320 -> 320, 8 x 64 x 64, GroupNorm + SiLU in the epilogue (EPI 6, 256 x 160 tiles)", conv_case(8, 64, 64, 320, 320, True), 256)
report("level-0 conv 640 -> 320 (EPI 6)", conv_case(8, 64, 64, 640, 320, True), 256)
report("level-0 conv 320 -> 320 + residual, raw output too (EPI 6)", conv_case(8, 64, 64, 320, 320, True, resid=True), 256)
report("level-1 conv 640 -> 640, 8 x 32 x 32 (EPI 6, 128 x 160 tiles)", conv_case(8, 32, 32, 640, 640, True), 256)
a = torch.randn(8192, 640, device=dev).bfloat16(); wt = (torch.randn(640, 640, device=dev) * 0.04).bfloat16(); bi = torch.randn(640, device=dev)
rs = torch.randn(8192, 640, device=dev).bfloat16()
report("plain GEMM 8192 x 640 x 640 + bias + residual (level-1 to_out shape)", lambda: ops.gemm(a, wt, bi, rs), 320)
a2 = torch.randn(8192, 2560, device=dev).bfloat16(); wt2 = (torch.randn(640, 2560, device=dev) * 0.02).bfloat16()
report("GEMM 8192 x 640 x 2560 + bias + residual (level-1 feed-forward output shape)", lambda: ops.gemm(a2, wt2, bi, rs), 256)
