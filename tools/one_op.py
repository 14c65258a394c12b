"""Run ONE operator shape repeatedly (for rocprofv3 --pmc / --kernel-trace on a single kernel).

  GILL_OP_REPEAT=20 python tools/one_op.py conv  B H W C1 C2 Cout [splitk]
  GILL_OP_REPEAT=20 python tools/one_op.py gemm  M N K [splitk]
  GILL_OP_REPEAT=20 python tools/one_op.py conv8 B H W C1 C2 Cout [splitk]   (fp8 operands)
  GILL_OP_REPEAT=20 python tools/one_op.py attn  B H nq nkv d
  GILL_OP_REPEAT=1  python tools/one_op.py geglu M inner K
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gill_amd
gill_amd.configure_hip_runtime()
from gill_amd import ops  # noqa: E402


def main():
  dev = torch.device("cuda:0")
  kind = sys.argv[1]
  a = [int(x) for x in sys.argv[2:]]
  if kind == "conv":
    B, H, W, C1, C2, Cout = a[:6]
    sk = a[6] if len(a) > 6 else 1
    x1 = torch.randn(B, H, W, C1, device=dev).bfloat16()
    x2 = torch.randn(B, H, W, C2, device=dev).bfloat16() if C2 else None
    w = torch.randn(Cout, C1 + C2, 3, 3, device=dev) * 0.02
    fn = lambda: ops.conv3x3(x1, w, x2=x2, splitk=sk)  # noqa: E731
    flops = 2.0 * B * H * W * Cout * 9 * (C1 + C2)
  elif kind == "conv8":
    B, H, W, C1, C2, Cout = a[:6]
    sk = a[6] if len(a) > 6 else 1
    x1 = torch.randn(B, H, W, C1 + C2, device=dev).bfloat16()
    w = torch.randn(Cout, C1 + C2, 3, 3, device=dev) * 0.02
    fn = lambda: ops.conv3x3_fp8(x1, w, splitk=sk)  # noqa: E731
    flops = 2.0 * B * H * W * Cout * 9 * (C1 + C2)
  elif kind == "gemm":
    M, N, K = a[:3]
    sk = a[3] if len(a) > 3 else 1
    x = torch.randn(M, K, device=dev).bfloat16()
    w = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    fn = lambda: ops.gemm(x, w, splitk=sk)  # noqa: E731
    flops = 2.0 * M * N * K
  elif kind == "geglu":
    M, inner, K = a[:3]
    x = torch.randn(M, K, device=dev).bfloat16()
    w = (torch.randn(2 * inner, K, device=dev) * 0.05).bfloat16()
    b = torch.randn(2 * inner, device=dev)
    fn = lambda: ops.geglu(x, w, b)  # noqa: E731
    flops = 2.0 * M * 2 * inner * K
  elif kind == "attn":
    B, H, nq, nkv, d = a[:5]
    q = torch.randn(B, nq, H * d, device=dev).bfloat16()
    k = torch.randn(B, nkv, H * d, device=dev).bfloat16()
    v = torch.randn(B, nkv, H * d, device=dev).bfloat16()
    fn = lambda: ops.attention(q, k, v, H)  # noqa: E731
    flops = 4.0 * B * H * nq * nkv * d
  else:
    raise SystemExit(__doc__)
  rep = int(os.environ.get("GILL_OP_REPEAT", "1"))
  fn()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record(); fn(); e1.record(); torch.cuda.synchronize()
  t = e0.elapsed_time(e1) * 1e-3 / rep
  print(f"{kind} {a}: {t * 1e6:.1f} us/launch (incl. wrapper overhead / {rep}), {flops / t / 1e12:.0f} TFLOP/s")


if __name__ == "__main__":
  main()
