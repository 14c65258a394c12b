"""Sweep GEMM / conv tile configurations at the SD-1.5 UNet shapes (batch Bx = 2 x prompts).

  python tools/sweep_gemm.py            # parent: spawns one child per (stages, bn) with the env knobs set
Each child times every shape for several split-K factors with GILL_OP_REPEAT launches per call, so the operator
wrapper's allocation / weight re-layout is amortised out.  Prints TFLOP/s (algorithmic).
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REPEAT = 50


def shapes(Bx):
  convs = [  # (tag, B, H, W, C1, C2, Cout, stride, ups)
    ("conv L0 320", Bx, 64, 64, 320, 0, 320, 1, 0),
    ("conv L1 640", Bx, 32, 32, 640, 0, 640, 1, 0),
    ("conv L2 1280", Bx, 16, 16, 1280, 0, 1280, 1, 0),
    ("conv L3 1280", Bx, 8, 8, 1280, 0, 1280, 1, 0),
    ("conv up L1 1280+640->640", Bx, 32, 32, 1280, 640, 640, 1, 0),
    ("conv up L0 640+320->320", Bx, 64, 64, 640, 320, 320, 1, 0),
    ("conv ups 16->32 1280", Bx, 16, 16, 1280, 0, 1280, 1, 1),
  ]
  gemms = [  # (tag, M, N, K)
    ("proj L0", Bx * 4096, 320, 320), ("ff2 L0", Bx * 4096, 320, 1280), ("qkv L0", Bx * 4096, 1152, 320),
    ("proj L1", Bx * 1024, 640, 640), ("ff2 L1", Bx * 1024, 640, 2560), ("qkv L1", Bx * 1024, 1920, 640),
    ("proj L2", Bx * 256, 1280, 1280), ("ff2 L2", Bx * 256, 1280, 5120),
  ]
  return convs, gemms


def child():
  import torch
  from gill_amd import ops
  dev = torch.device("cuda:0")
  Bx = int(os.environ.get("BX", "8"))
  convs, gemms = shapes(Bx)
  res = []

  def timeit(fn):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
      e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      e0.record(); fn(); e1.record(); torch.cuda.synchronize()
      best = min(best, e0.elapsed_time(e1))
    return best * 1e-3 / REPEAT   # wrapper overhead (one alloc + re-layout) is amortised over REPEAT launches

  for (tag, B, H, W, C1, C2, Cout, stride, ups) in convs:
    x1 = torch.randn(B, H, W, C1, device=dev).bfloat16()
    x2 = torch.randn(B, H, W, C2, device=dev).bfloat16() if C2 else None
    w = torch.randn(Cout, C1 + C2, 3, 3, device=dev) * 0.02
    OH = 2 * H if ups else H
    fl = 2.0 * B * OH * OH * Cout * 9 * (C1 + C2)
    for sk in (1, 2, 3, 4, 6, 8, 12):
      if sk > 1 and B * OH * OH * Cout > 8 * 4096 * 640:
        continue
      t = timeit(lambda: ops.conv3x3(x1, w, x2=x2, upsample=bool(ups), splitk=sk))
      res.append(dict(tag=tag, sk=sk, us=t * 1e6, tflops=fl / t / 1e12))
  for (tag, M, N, K) in gemms:
    a = torch.randn(M, K, device=dev).bfloat16()
    w = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    for sk in (1, 2, 4):
      if sk > 1 and M * N > 2048 * 1280:
        continue
      t = timeit(lambda: ops.gemm(a, w, splitk=sk))
      res.append(dict(tag=tag, sk=sk, us=t * 1e6, tflops=2.0 * M * N * K / t / 1e12))
  print("RESULT " + json.dumps(res), flush=True)


def main():
  if os.environ.get("SWEEP_CHILD") == "1":
    return child()
  table = {}
  for stages in (2, 3):
    for bn in (128, 160):
      env = dict(os.environ, SWEEP_CHILD="1", GILL_GEMM_STAGES=str(stages), GILL_GEMM_BN=str(bn), GILL_OP_REPEAT=str(REPEAT))
      out = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True)
      line = [l for l in out.stdout.splitlines() if l.startswith("RESULT ")]
      if not line:
        print(f"stages={stages} bn={bn} FAILED:\n{out.stdout[-2000:]}\n{out.stderr[-2000:]}")
        continue
      for r in json.loads(line[0][7:]):
        table.setdefault((r["tag"], r["sk"]), {})[(stages, bn)] = (r["us"], r["tflops"])
  cfgs = [(2, 128), (2, 160), (3, 128), (3, 160)]
  print("| shape | splitk | " + " | ".join(f"st{s} bn{b} us / TF" for s, b in cfgs) + " |")
  print("|---|---|" + "---|" * len(cfgs))
  for (tag, sk), v in table.items():
    cells = []
    for c in cfgs:
      cells.append(f"{v[c][0]:.1f} / {v[c][1]:.0f}" if c in v else "-")
    print(f"| {tag} | {sk} | " + " | ".join(cells) + " |")


if __name__ == "__main__":
  main()
