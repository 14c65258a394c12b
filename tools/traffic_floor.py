#!/usr/bin/env python
"""Algorithmic HBM bytes of ONE SD-1.5 UNet forward in this build's dataflow: the bf16 weights once, plus one
read per operand and one write per result of every kernel whose output is a materialised tensor (bf16 NHWC activations;
attention operands with their head padding; fp32 latents).  Nothing is counted for operands a kernel re-reads from cache
(3x3 taps, K/V tiles), nothing for split-K partials or statistics: this is the floor the measured PMC traffic is compared with.
  python tools/traffic_floor.py [batch]"""
import sys

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
ch = (320, 640, 1280, 1280)
L = 64
heads = 8
bf = 2
tot_r = tot_w = 0
rows = []


def add(name, r, w):
  global tot_r, tot_w
  tot_r += r
  tot_w += w
  rows.append((name, r, w))


def resnet(tag, hw, cin, cout):
  px = B * hw * bf
  # GN-apply(x)->n1 ; conv1(n1)->h ; GN-apply(h)->n2 ; conv2(n2 [+ x via shortcut / residual])->out
  add(tag + " resnet", px * (cin + cin + cout + cout + cin), px * (cin + cout + cout + cout))


def xf(tag, hw, c):
  px = B * hw * bf
  d = c // heads
  dp = 48 if d <= 48 else (64 if d <= 64 else (80 if d <= 80 else (128 if d <= 128 else 160)))
  hq = heads * dp
  dpv = (dp + 31) // 32 * 32
  r = c; w = c                          # GN-apply: x -> n
  if c == 320:
    # level 0: the row-resident kernels (lnproj.hip, ffn.hip) keep t's LayerNorm-ed copy, the GEGLU product, the feed-forward output
    # and the attn2.to_out result in registers
    r += c; w += c + 2 * hq + heads * dpv                 # lnproj 0: n -> t, q, k, vt
    r += 2 * hq + heads * dpv; w += hq                    # self-attention
    r += hq + c; w += c + hq                              # lnproj 1: o, t -> t, q
    r += hq; w += hq                                      # cross-attention (K/V of 77 tokens: negligible)
    r += hq + c + c; w += c                               # ffn (PRE form): o, t, outer residual x -> out
  else:
    r += c; w += c                                        # proj_in: n -> t
    r += c; w += 2 * hq + heads * dpv                     # QKV: reads t, writes q, k, vt
    r += 2 * hq + heads * dpv; w += hq                    # self-attention
    r += hq + c; w += c                                   # out1 (+ residual t)
    # cross-attention as two GEMMs on per-sample folded weights (unet.hip "XALG"): P = softmax80(LN(t) Mq^T) [80 x heads wide], t += P Wo^T
    # (+ per call and sample the operands Mq, Wo: 2 x 80 x heads x c bf16 elements, added below as "weights")
    r += c; w += 80 * heads                               # scores + softmax GEMM: reads t, writes P
    r += 80 * heads + c; w += c                           # values + to_out GEMM (+ residual t)
    r += c; w += 4 * c                                    # GEGLU
    r += 4 * c + c + c; w += c                            # [Wp W2 | Wp] GEMM over [h | t] + outer residual x -> out
  add(tag + " transformer", px * r, px * w)


hw = L * L
add("conv_in (im2col + GEMM)", B * hw * (4 * 4 + 64 * bf), B * hw * (64 * bf + ch[0] * bf))
c_prev = ch[0]
skips = [ch[0]]
for i in range(4):
  for j in range(2):
    resnet(f"down{i}.{j}", hw, c_prev if j == 0 else ch[i], ch[i])
    if i < 3:
      xf(f"down{i}.{j}", hw, ch[i])
    skips.append(ch[i])
    c_prev = ch[i]
  if i < 3:
    add(f"down{i} downsample conv", B * hw * ch[i] * bf, B * hw // 4 * ch[i] * bf)
    hw //= 4
    skips.append(ch[i])
resnet("mid.0", hw, ch[3], ch[3]); xf("mid", hw, ch[3]); resnet("mid.1", hw, ch[3], ch[3])
rev = ch[::-1]
for i in range(4):
  for j in range(3):
    resnet(f"up{i}.{j}", hw, c_prev + skips.pop(), rev[i])
    c_prev = rev[i]
    if i > 0:
      xf(f"up{i}.{j}", hw, rev[i])
  if i < 3:
    add(f"up{i} upsample conv", B * hw * rev[i] * bf, B * hw * 4 * rev[i] * bf)
    hw *= 4
add("conv_norm_out + conv_out", B * hw * ch[0] * bf * 2, B * hw * (ch[0] * bf + 4 * 4))
weights = 859_520_964 * bf
# XALG layers (5 at C = 640, 6 at C = 1280) read per-sample operands instead of to_q / to_out weights: 2 x 80 x heads x C per sample
weights += sum(n * (B * 2 * 80 * heads * c - 2 * c * c) * bf for n, c in ((5, 640), (6, 1280)))
act = tot_r + tot_w
print(f"SD-1.5 UNet forward, batch {B}: weights {weights / 1e9:.2f} GB + activations {tot_r / 1e9:.2f} GB read + {tot_w / 1e9:.2f} GB written "
      f"= {(weights + act) / 1e9:.2f} GB algorithmic HBM bytes")
for name, r, w in sorted(rows, key=lambda t: -(t[1] + t[2]))[:8]:
  print(f"  {name:28s} {r / 1e6:8.1f} MB read {w / 1e6:8.1f} MB written")
