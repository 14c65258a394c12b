import sys, os, torch
sys.path.insert(0, "/root/repo")
from gill_amd import ops
dev = torch.device("cuda:0")
for (B, H, C1, C2) in [(8, 64, 320, 0), (8, 64, 640, 320), (8, 64, 320, 320), (8, 32, 640, 0), (8, 32, 1280, 640), (8, 16, 1280, 0), (8, 16, 1280, 1280), (8, 8, 1280, 1280)]:
  x1 = torch.randn(B, H, H, C1, device=dev).bfloat16()
  x2 = torch.randn(B, H, H, C2, device=dev).bfloat16() if C2 else None
  g = torch.ones(C1 + C2, device=dev); b = torch.zeros(C1 + C2, device=dev)
  for _ in range(5):
    y = ops.groupnorm(x1, g, b, 32, 1e-5, True, x2)
  torch.cuda.synchronize()
print("done")
