"""How much slower is a GEMM / conv whose WEIGHTS are HBM-cold (as in the UNet loop: 1.7 GB of weights cycle through a 256 MB
Infinity Cache every forward) than the same launch with warm weights?  Run under rocprofv3 --kernel-trace --stats and compare the
kernel's average duration between modes:   python tools/cold_weights.py warm|coldw|coldall  gemm M N K | conv B H W Cin Cout"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gill_amd import ops

mode, kind = sys.argv[1], sys.argv[2]
a = [int(v) for v in sys.argv[3:]]
dev = torch.device("cuda:0")
evict = torch.empty(768 << 20, dtype=torch.uint8, device=dev)      # 3x the Infinity Cache
if kind == "gemm":
  M, N, K = a
  x = torch.randn(M, K, device=dev).bfloat16()
  w = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
  fn = lambda: ops.gemm(x, w, splitk=1)
else:
  B, H, W, Cin, Cout = a
  x = torch.randn(B, H, W, Cin, device=dev).bfloat16()
  w = torch.randn(Cout, Cin, 3, 3, device=dev) * 0.02
  fn = lambda: ops.conv3x3(x, w)
for it in range(12):
  if mode != "warm":
    evict.fill_(it)                       # everything out of L2 / Infinity Cache
    if mode == "coldw":
      _ = (x.float().sum())               # the activation back in (read once), the weights stay cold
  else:
    fn()
  torch.cuda.synchronize()
  fn()
  torch.cuda.synchronize()
