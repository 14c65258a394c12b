"""Micro-benchmarks of the libgill_amd kernels at the shapes the SD-1.5 UNet / OPT-6.7b use.
Prints achieved TFLOP/s (algorithmic FLOPs / wall time measured with HIP events on the launch stream)."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gill_amd import _native as N  # noqa: E402
from gill_amd import ops  # noqa: E402


def timeit(fn, iters=10, warm=3):
  for _ in range(warm):
    fn()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(iters):
    fn()
  e1.record()
  torch.cuda.synchronize()
  return e0.elapsed_time(e1) / iters * 1e-3


def main():
  dev = torch.device("cuda:0")
  res = []
  Bx = int(os.environ.get("BX", "8"))
  # GEMMs (M, N, K)
  for (M, N, K, tag) in [(Bx * 4096, 320, 320, "proj L0"), (Bx * 4096, 2560, 320, "ff1 L0 (as plain)"),
                         (Bx * 4096, 320, 1280, "ff2 L0"), (Bx * 1024, 640, 2560, "ff2 L1"),
                         (Bx * 256, 1280, 5120, "ff2 L2"), (4096, 4096, 4096, "square 4k"),
                         (8192, 8192, 8192, "square 8k"), (128, 16384, 4096, "opt fc1 M128"),
                         (128, 4096, 16384, "opt fc2 M128")]:
    a = torch.randn(M, K, device=dev).bfloat16()
    w = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    t = timeit(lambda: ops.gemm(a, w, splitk=0))
    res.append(dict(op="gemm", tag=tag, M=M, N=N, K=K, ms=t * 1e3, tflops=2.0 * M * N * K / t / 1e12))
    print(res[-1], flush=True)
  # convs: the per-call wrapper re-lays the weights each call, so time the raw kernel through a cached handle instead
  for (B, H, W, C1, C2, Cout, tag) in [(Bx, 64, 64, 320, 0, 320, "conv L0"), (Bx, 32, 32, 640, 0, 640, "conv L1"),
                                       (Bx, 16, 16, 1280, 0, 1280, "conv L2"), (Bx, 8, 8, 1280, 0, 1280, "conv L3"),
                                       (Bx, 16, 16, 1280, 1280, 1280, "conv up L2 cat")]:
    x1 = torch.randn(B, H, W, C1, device=dev).bfloat16()
    x2 = torch.randn(B, H, W, C2, device=dev).bfloat16() if C2 else None
    w = torch.randn(Cout, C1 + C2, 3, 3, device=dev) * 0.05
    t = timeit(lambda: ops.conv3x3(x1, w, x2=x2, splitk=0), iters=5, warm=2)
    fl = 2.0 * B * H * W * Cout * 9 * (C1 + C2)
    res.append(dict(op="conv3x3(+relayout+sync)", tag=tag, ms=t * 1e3, tflops=fl / t / 1e12))
    print(res[-1], flush=True)
  for (B, H, n, nkv, d, tag) in [(Bx, 8, 4096, 4096, 40, "self L0"), (Bx, 8, 1024, 1024, 80, "self L1"),
                                 (Bx, 8, 256, 256, 160, "self L2"), (Bx, 8, 4096, 77, 40, "cross L0")]:
    q = torch.randn(B, n, H * d, device=dev).bfloat16()
    k = torch.randn(B, nkv, H * d, device=dev).bfloat16()
    v = torch.randn(B, nkv, H * d, device=dev).bfloat16()
    t = timeit(lambda: ops.attention(q, k, v, H), iters=5, warm=2)
    fl = 4.0 * B * H * n * nkv * d
    res.append(dict(op="attention(+pack+sync)", tag=tag, ms=t * 1e3, tflops=fl / t / 1e12))
    print(res[-1], flush=True)
  os.makedirs("gpurun_out", exist_ok=True)
  with open("gpurun_out/bench_ops.json", "w") as f:
    json.dump(res, f, indent=1)


if __name__ == "__main__":
  main()
