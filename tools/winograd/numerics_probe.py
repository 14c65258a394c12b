"""Go / no-go probe for VERDICT r04 item 2 (Winograd F(2x2, 3x3) for the stride-1 3x3 convolutions), numerics half, CPU only.

The HIP path would run the 16 per-position GEMMs on bf16 operands: the transformed input V = B^T d B (sums / differences of four bf16
activations, rounded to bf16) and the transformed weights U = G g G^T (computed in fp32 at load, rounded to bf16), fp32 accumulation,
output transform A^T M A in fp32.  This script measures what that costs on a FULL-SIZE SD-1.5 UNet forward of the oracle
(oracle/unet_ref.py, fp32), one CFG pair, by swapping the oracle's conv for

  mode direct : conv(bf16(x), bf16(w)) in fp32                   -- what the shipped implicit-GEMM conv computes
  mode wino   : Winograd with bf16(V), bf16(U), fp32 accumulate  -- the candidate

on the convolutions selected by --cin-min (all other arithmetic stays fp32), and printing the relative L2 distance to the plain fp32
forward.  Usage: python tools/winograd/numerics_probe.py [--cin-min 640] [--tiny]
"""
import argparse
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gill_amd import synth          # noqa: E402
from oracle import unet_ref         # noqa: E402

BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float32)
G = torch.tensor([[1, 0, 0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0, 0, 1]], dtype=torch.float32)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float32)


def bf(x):
  return x.bfloat16().float()


def winograd_conv(x, w, b):
  """x (B, C, H, W) fp32, w (O, C, 3, 3), padding 1, stride 1; H, W even."""
  Bn, C, H, W = x.shape
  O = w.shape[0]
  xp = F.pad(bf(x), (1, 1, 1, 1))
  # 4x4 patches with stride 2: (B, C, H/2, W/2, 4, 4)
  d = xp.unfold(2, 4, 2).unfold(3, 4, 2)
  V = torch.einsum("ij,bchwjk,lk->bchwil", BT, d, BT)
  V = bf(V)
  U = bf(torch.einsum("ij,ocjk,lk->ocil", G, bf(w), G))
  M = torch.einsum("bchwil,ocil->bohwil", V, U)
  Y = torch.einsum("ij,bohwjk,lk->bohwil", AT, M, AT)          # (B, O, H/2, W/2, 2, 2)
  Y = Y.permute(0, 1, 2, 4, 3, 5).reshape(Bn, O, H, W)
  return Y + b.view(1, -1, 1, 1)


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--cin-min", type=int, default=640)
  ap.add_argument("--tiny", action="store_true")
  ap.add_argument("--threads", type=int, default=0)
  args = ap.parse_args()
  if args.threads:
    torch.set_num_threads(args.threads)
  cfg = synth.UNetConfig.tiny(16) if args.tiny else synth.UNetConfig.sd15()
  sd = {k: v.bfloat16().float() for k, v in synth.unet_state_dict(cfg, seed=51).items()}
  L = cfg.sample_size
  x = synth.initial_latents(2, 4, L, seed=4242)
  ctx = synth.normal("wino_ctx", (2, cfg.ctx_len, cfg.cross_attention_dim), 7).bfloat16().float()
  t = torch.tensor([501.0, 501.0])
  orig = unet_ref._conv
  stats = {"n": 0}

  def run(mode):
    def conv(sdd, p, xx, stride=1, padding=1):
      w, b = sdd[p + ".weight"].float(), sdd[p + ".bias"].float()
      sel = (w.shape[2] == 3 and stride == 1 and w.shape[1] >= args.cin_min and xx.shape[2] % 2 == 0 and xx.shape[2] >= 16 - (8 if args.tiny else 0))
      if mode == "fp32" or not sel:
        return orig(sdd, p, xx, stride, padding)
      stats["n"] += 1
      if mode == "direct":
        return F.conv2d(bf(xx), bf(w), b, stride=1, padding=1)
      return winograd_conv(xx, w, b)
    unet_ref._conv = conv
    t0 = time.time()
    with torch.no_grad():
      y = unet_ref.unet_forward(sd, x, t, ctx, cfg.block_out_channels, cfg.num_heads, cfg.norm_num_groups)
    unet_ref._conv = orig
    return y, time.time() - t0

  ref, dt = run("fp32")
  print(f"fp32 forward {dt:.1f} s")
  for mode in ("direct", "wino"):
    stats["n"] = 0
    y, dt = run(mode)
    rel = ((y - ref).norm() / ref.norm()).item()
    print(f"mode {mode:6s}: {stats['n']} convolutions replaced (Cin >= {args.cin_min}, maps >= 16x16), rel-L2 vs fp32 {rel:.3e}  ({dt:.1f} s)")


if __name__ == "__main__":
  main()
