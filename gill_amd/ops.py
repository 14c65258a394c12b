"""Operator-level Python wrappers over libgill_amd (gill_op_* in include/gill_amd.h).

These exist so the parity tests can pin every kernel the three stages are built from; the stages
themselves (gill_amd.models / layers / sd) call the stage-level entry points, not these.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import _native as N

ACT = {"none": 0, "relu": 1, "gelu": 2, "silu": 3}


def _bf(t: torch.Tensor) -> torch.Tensor:
  assert t.dtype == torch.bfloat16 and t.is_cuda
  return t.contiguous()


def gemm(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, resid: Optional[torch.Tensor] = None,
         alpha: float = 1.0, act: str = "none", out_f32: bool = False, splitk: int = 0, row_major: bool = False) -> torch.Tensor:
  """act(alpha * a @ w.T + bias + resid); a (M,K) bf16, w (N,K) bf16, bias (N) fp32, resid (M,N) bf16.
  row_major: keep weight-streaming shapes (N * K >= 4 Mi) on row-major weights instead of the 64 x 64-blocked copy (STREAM64 tile up to 256 rows, general
  tiles on the blocked layout above)."""
  if row_major:
    splitk = -1 if splitk <= 1 else -splitk
  a, w = _bf(a), _bf(w)
  M, K = a.shape
  Nn = w.shape[0]
  out = torch.empty((M, Nn), device=a.device, dtype=torch.float32 if out_f32 else torch.bfloat16)
  if bias is not None:
    bias = bias.float().contiguous()
  if resid is not None:
    resid = _bf(resid)
  N.check(N.lib().gill_op_gemm(N.ptr(a), N.ptr(w), N.ptr(bias), N.ptr(resid), N.ptr(out), M, Nn, K, alpha, ACT[act],
                               int(out_f32), splitk, N.current_stream()))
  return out


def geglu_fp8(t: torch.Tensor, ln_g: torch.Tensor, ln_b: torch.Tensor, w: torch.Tensor, bias: torch.Tensor) -> torch.Tensor:
  """BASELINE configs[4]: diffusers GEGLU of LayerNorm(t) on the fp8 matrix instruction (csrc/linear_fp8.hip): t (M,C) bf16, w (2 inner, C) bf16 in
  diffusers order, bias (2 inner); activations and weights quantised to e4m3 as the engine's fp8 mode does.  Returns (M, inner) bf16."""
  t, w = _bf(t), _bf(w)
  M, Cc = t.shape
  inner = w.shape[0] // 2
  out = torch.empty((M, inner), device=t.device, dtype=torch.bfloat16)
  f = lambda x: x.float().contiguous()   # noqa: E731
  ln_g, ln_b, bias = f(ln_g), f(ln_b), f(bias)
  N.check(N.lib().gill_op_geglu_fp8(N.ptr(t), N.ptr(ln_g), N.ptr(ln_b), N.ptr(w), N.ptr(bias), N.ptr(out), M, inner, Cc, N.current_stream()))
  return out


def geglu(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor]) -> torch.Tensor:
  """diffusers GEGLU: h, g = (a @ w.T + bias).chunk(2, -1); h * gelu(g)."""
  a, w = _bf(a), _bf(w)
  M, K = a.shape
  inner = w.shape[0] // 2
  out = torch.empty((M, inner), device=a.device, dtype=torch.bfloat16)
  if bias is not None:
    bias = bias.float().contiguous()
  N.check(N.lib().gill_op_geglu(N.ptr(a), N.ptr(w), N.ptr(bias), N.ptr(out), M, inner, K, N.current_stream()))
  return out


def conv3x3(x1: torch.Tensor, w_oihw: torch.Tensor, bias: Optional[torch.Tensor] = None, x2: Optional[torch.Tensor] = None,
            rowvec: Optional[torch.Tensor] = None, resid: Optional[torch.Tensor] = None, stride: int = 1,
            upsample: bool = False, splitk: int = 0) -> torch.Tensor:
  """3x3 / pad 1 conv over NHWC bf16 x1 (B,H,W,C1) [channel-concat x2], weights OIHW fp32 -> NHWC bf16."""
  x1 = _bf(x1)
  B, IH, IW, C1 = x1.shape
  C2 = 0
  if x2 is not None:
    x2 = _bf(x2)
    C2 = x2.shape[-1]
  w = w_oihw.float().contiguous()
  Cout = w.shape[0]
  assert w.shape[1] == C1 + C2
  if upsample:
    OH, OW = 2 * IH, 2 * IW
  else:
    OH, OW = (IH + 2 - 3) // stride + 1, (IW + 2 - 3) // stride + 1
  y = torch.empty((B, OH, OW, Cout), device=x1.device, dtype=torch.bfloat16)
  if bias is not None:
    bias = bias.float().contiguous()
  if rowvec is not None:
    rowvec = rowvec.float().contiguous()
  if resid is not None:
    resid = _bf(resid)
  N.check(N.lib().gill_op_conv3x3(N.ptr(x1), C1, N.ptr(x2), C2, N.ptr(w), N.ptr(bias), N.ptr(rowvec), N.ptr(resid),
                                  N.ptr(y), B, IH, IW, Cout, stride, int(upsample), splitk, N.current_stream()))
  return y


def conv3x3_gn(x: torch.Tensor, w_oihw: torch.Tensor, bias: Optional[torch.Tensor], gamma: torch.Tensor, beta: torch.Tensor,
               groups: int = 32, eps: float = 1e-5, silu: bool = True, resid: Optional[torch.Tensor] = None, splitk: int = 2,
               want_raw: bool = True, coop: bool = True, rowvec: Optional[torch.Tensor] = None, want_table: bool = False, want_norm: bool = True):
  """3x3 convolution + the consuming GroupNorm (+ SiLU) without a GroupNorm launch of its own (include/gill_amd.h gill_op_conv3x3_gn):
  splitk >= 2 in the split-K reducer (coop=False) or inside the convolution's launch (coop=True, where the geometry allows); splitk == 1 in the
  convolution's epilogue (coop=True) or as conv + GroupNorm-apply (coop=False, the reference dataflow).  x (B,H,W,Cin) bf16 NHWC, rowvec (B,Cout).
  Returns (y_raw or None, y_norm or None[, table (B,2,Cout) fp32 when want_table]) — outputs NaN-prefilled, so an unwritten element shows."""
  x = _bf(x)
  B, H, W, Cin = x.shape
  Cout = w_oihw.shape[0]
  nan = float("nan")
  y_raw = torch.full((B, H, W, Cout), nan, device=x.device, dtype=torch.bfloat16) if want_raw else None
  y_norm = torch.full((B, H, W, Cout), nan, device=x.device, dtype=torch.bfloat16) if want_norm else None
  table = torch.full((B, 2, Cout), nan, device=x.device, dtype=torch.float32) if want_table else None
  f = lambda t: None if t is None else t.float().contiguous()   # noqa: E731
  w, bias, gamma, beta, rowvec = f(w_oihw), f(bias), f(gamma), f(beta), f(rowvec)
  resid = None if resid is None else _bf(resid)
  N.check(N.lib().gill_op_conv3x3_gn(N.ptr(x), N.ptr(w), N.ptr(bias), N.ptr(rowvec), N.ptr(resid), N.ptr(gamma), N.ptr(beta), groups, float(eps),
                                     int(silu), N.ptr(y_raw), N.ptr(y_norm), N.ptr(table), B, H, W, Cin, Cout, splitk, int(coop), N.current_stream()))
  return (y_raw, y_norm, table) if want_table else (y_raw, y_norm)


def conv3x3_shortcut(x1: torch.Tensor, w_oihw: torch.Tensor, xs1: torch.Tensor, w_sc: torch.Tensor, bias: Optional[torch.Tensor] = None,
                     x2: Optional[torch.Tensor] = None, xs2: Optional[torch.Tensor] = None, splitk: int = 0) -> torch.Tensor:
  """conv3x3(x1 ++ x2, w_oihw) + bias + conv1x1(xs1 ++ xs2, w_sc) as one implicit GEMM (ResnetBlock2D conv2 + conv_shortcut)."""
  x1, xs1 = _bf(x1), _bf(xs1)
  B, IH, IW, C1 = x1.shape
  C2 = CS2 = 0
  if x2 is not None:
    x2 = _bf(x2); C2 = x2.shape[-1]
  if xs2 is not None:
    xs2 = _bf(xs2); CS2 = xs2.shape[-1]
  CS1 = xs1.shape[-1]
  w, wsc = w_oihw.float().contiguous(), w_sc.float().contiguous()
  Cout = w.shape[0]
  assert w.shape[1] == C1 + C2 and tuple(wsc.shape) == (Cout, CS1 + CS2)
  y = torch.empty((B, IH, IW, Cout), device=x1.device, dtype=torch.bfloat16)
  if bias is not None:
    bias = bias.float().contiguous()
  N.check(N.lib().gill_op_conv3x3_shortcut(N.ptr(x1), C1, N.ptr(x2), C2, N.ptr(w), N.ptr(bias), N.ptr(xs1), CS1, N.ptr(xs2), CS2,
                                           N.ptr(wsc), N.ptr(y), B, IH, IW, Cout, splitk, N.current_stream()))
  return y


def attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, heads: int, scale: Optional[float] = None,
              causal: bool = False) -> torch.Tensor:
  """q (B,nq,H*d), k/v (B,nkv,H*d) bf16 -> (B,nq,H*d) bf16."""
  q, k, v = _bf(q), _bf(k), _bf(v)
  B, nq, hd = q.shape
  nkv = k.shape[1]
  d = hd // heads
  if scale is None:
    scale = d ** -0.5
  o = torch.empty_like(q)
  N.check(N.lib().gill_op_attention(N.ptr(q), N.ptr(k), N.ptr(v), N.ptr(o), B, heads, nq, nkv, d, float(scale),
                                    int(causal), N.current_stream()))
  return o


def layernorm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
  assert x.is_cuda and x.dtype in (torch.bfloat16, torch.float32)
  x = x.contiguous()
  Cc = x.shape[-1]
  rows = x.numel() // Cc
  y = torch.empty(x.shape, device=x.device, dtype=torch.bfloat16)
  g, b = gamma.float().contiguous(), beta.float().contiguous()
  N.check(N.lib().gill_op_layernorm(N.ptr(x), int(x.dtype == torch.float32), N.ptr(g), N.ptr(b), N.ptr(y), rows, Cc, eps,
                                    N.current_stream()))
  return y


def groupnorm(x1: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, groups: int = 32, eps: float = 1e-5,
              silu: bool = False, x2: Optional[torch.Tensor] = None) -> torch.Tensor:
  """GroupNorm(+SiLU) over NHWC bf16 (B,H,W,C1) [channel-concat x2] -> NHWC bf16 (B,H,W,C1+C2)."""
  x1 = _bf(x1)
  B, H, W, C1 = x1.shape
  C2 = 0
  if x2 is not None:
    x2 = _bf(x2)
    C2 = x2.shape[-1]
  y = torch.empty((B, H, W, C1 + C2), device=x1.device, dtype=torch.bfloat16)
  g, b = gamma.float().contiguous(), beta.float().contiguous()
  N.check(N.lib().gill_op_groupnorm(N.ptr(x1), C1, N.ptr(x2), C2, B, H * W, groups, N.ptr(g), N.ptr(b), eps, int(silu),
                                    N.ptr(y), N.current_stream()))
  return y


def conv3x3_fp8(x: torch.Tensor, w_oihw: torch.Tensor, bias: Optional[torch.Tensor] = None, resid: Optional[torch.Tensor] = None,
                splitk: int = 0) -> torch.Tensor:
  """3x3 / pad 1 / stride 1 conv with fp8 (e4m3) activations and weights on the fp8 MFMA (csrc/conv_fp8.hip): NHWC bf16 x
  (B,H,W,Cin), OIHW fp32 weights -> NHWC bf16.  Quantisation (x * 8 per tensor, weights per output channel) happens inside."""
  x = _bf(x)
  B, H, W, Cin = x.shape
  w = w_oihw.float().contiguous()
  Cout = w.shape[0]
  assert w.shape[1] == Cin and Cin % 64 == 0
  y = torch.empty((B, H, W, Cout), device=x.device, dtype=torch.bfloat16)
  if bias is not None:
    bias = bias.float().contiguous()
  if resid is not None:
    resid = _bf(resid)
  N.check(N.lib().gill_op_conv3x3_fp8(N.ptr(x), N.ptr(w), N.ptr(bias), N.ptr(resid), N.ptr(y), B, H, W, Cin, Cout, splitk,
                                      N.current_stream()))
  return y


def ffn_fused(t: torch.Tensor, ln_g: torch.Tensor, ln_b: torch.Tensor, w1: torch.Tensor, b1: torch.Tensor, w2: torch.Tensor,
              b2: torch.Tensor, wp: torch.Tensor, bp: torch.Tensor, resid: torch.Tensor, rows_per_batch: int = 0,
              o2: Optional[torch.Tensor] = None, wo: Optional[torch.Tensor] = None, bo2: Optional[torch.Tensor] = None):
  """The feed-forward sub-block of a C = 320 transformer block + proj_out + outer residual as one kernel (csrc/ffn.hip):
  out = (t + ff2(value * gelu(gate))) @ wp.T + bp + resid with [value | gate] = LN(t) @ w1.T + b1 (diffusers layouts: w1 (2560, 320),
  w2 (320, 1280), wp (320, 320)).  t, resid (M, 320) bf16, M % 128 == 0.  Returns out (M, 320) bf16 [, GroupNorm partial sums
  (M / 64, 64, 2) fp32 when rows_per_batch > 0].  o2 / wo / bo2 (all or none): t := t + o2 @ wo.T + bo2 first, inside the kernel
  (BasicTransformerBlock.attn2.to_out + its residual; o2 (M, 320), wo (320, 320)) — the form the UNet engine runs."""
  t, w1, w2, wp, resid = (_bf(x) for x in (t, w1, w2, wp, resid))
  M = t.shape[0]
  # outputs start as NaN: the engine hands these kernels stale arena memory, so a tile the kernel forgot to write must show in a test
  out = torch.full((M, 320), float("nan"), device=t.device, dtype=torch.bfloat16)
  stats = torch.full((M // 64, 64, 2), float("nan"), device=t.device, dtype=torch.float32) if rows_per_batch else None
  f = lambda x: x.float().contiguous()
  ln_g, ln_b, b1, b2, bp = f(ln_g), f(ln_b), f(b1), f(b2), f(bp)
  if o2 is not None:       # the PRE form: t := t + o2 @ wo.T + bo2 first, inside the kernel
    o2, wo, bo2 = _bf(o2), _bf(wo), f(bo2)
  N.check(N.lib().gill_op_ffn_fused(N.ptr(t), N.ptr(ln_g), N.ptr(ln_b), N.ptr(w1), N.ptr(b1), N.ptr(w2), N.ptr(b2), N.ptr(wp), N.ptr(bp),
                                    N.ptr(resid), N.ptr(out), N.ptr(stats), M, rows_per_batch, N.ptr(o2), N.ptr(wo), N.ptr(bo2),
                                    N.current_stream()))
  return (out, stats) if rows_per_batch else out


def lnproj(mode: int, x: torch.Tensor, t, w1: torch.Tensor, b1: torch.Tensor, ln_g: torch.Tensor, ln_b: torch.Tensor, w2: torch.Tensor,
           B: int, HW: int):
  """The two projections around norm1 (mode 0) / norm2 (mode 1) of a C = 320, 8-head transformer block as one kernel (csrc/lnproj.hip).
  mode 0: t = x @ w1.T + b1; q, k, v = LN(t) @ w2.T with w2 (960, 320) = to_q | to_k | to_v.  mode 1: t = t + x @ w1.T + b1 (x = the
  attention output); q = LN(t) @ w2.T with w2 (320, 320).  Returns (t (M, 320) bf16, q, k, vt) in the attention kernels' layouts
  (q, k (B, 8, hw_pad, 48), q pre-scaled by log2(e) / sqrt(40); vt (B, 8, 64, hw_pad) with row 48 = 1); k, vt None in mode 1."""
  x, w1, w2 = (_bf(v) for v in (x, w1, w2))
  M = B * HW
  hw_pad = (HW + 31) // 32 * 32
  # outputs start as NaN (the engine passes stale arena memory): pad columns 40..47, the ones-row of V^T and every row must be WRITTEN
  nan = float("nan")
  t = torch.full((M, 320), nan, device=x.device, dtype=torch.bfloat16) if mode == 0 else _bf(t).clone()
  q = torch.full((B, 8, hw_pad, 48), nan, device=x.device, dtype=torch.bfloat16)
  k = torch.full_like(q, nan) if mode == 0 else None
  vt = torch.full((B, 8, 64, hw_pad), nan, device=x.device, dtype=torch.bfloat16) if mode == 0 else None
  f = lambda v: v.float().contiguous()
  b1, ln_g, ln_b = f(b1), f(ln_g), f(ln_b)
  N.check(N.lib().gill_op_lnproj(mode, N.ptr(x), N.ptr(t), N.ptr(w1), N.ptr(b1), N.ptr(ln_g), N.ptr(ln_b), N.ptr(w2), N.ptr(q), N.ptr(k),
                                 N.ptr(vt), B, HW, N.current_stream()))
  return t, q, k, vt


def cross_attention_folded(t: torch.Tensor, ln_g: torch.Tensor, ln_b: torch.Tensor, wq: torch.Tensor, wk: torch.Tensor, wv: torch.Tensor,
                           wo: torch.Tensor, bo: torch.Tensor, ctx: torch.Tensor, heads: int, B: int, HW: int):
  """norm2 + attn2 + residual of a BasicTransformerBlock with heads of 80..160 features as the engine runs it at UNet levels 1-3
  (csrc/unet.hip "XALG"): out = t + softmax(LN(t) wq.T (ctx wk.T).T / sqrt(d)) (ctx wv.T) wo.T + bo, computed as two GEMMs on per-sample
  weights folded from (wq, wk) and (wo, wv).  t (B * HW, C); wq, wo (C, C); wk, wv (C, E); ctx (B, ctx_len <= 80, E).
  Returns (out (B * HW, C) bf16, P (B * HW, 80 * heads) bf16: the softmax weights, key j of head h at column 80 h + j)."""
  t, wq, wk, wv, wo, ctx = (_bf(v) for v in (t, wq, wk, wv, wo, ctx))
  M, C = t.shape
  E = ctx.shape[-1]
  f = lambda v: v.float().contiguous()
  ln_g, ln_b, bo = f(ln_g), f(ln_b), f(bo)
  nan = float("nan")      # (the engine passes stale arena memory: every element must be written)
  out = torch.full((M, C), nan, device=t.device, dtype=torch.bfloat16)
  P = torch.full((M, 80 * heads), nan, device=t.device, dtype=torch.bfloat16)
  N.check(N.lib().gill_op_cross_attention_folded(N.ptr(t), N.ptr(ln_g), N.ptr(ln_b), N.ptr(wq), N.ptr(wk), N.ptr(wv), N.ptr(wo), N.ptr(bo),
                                                 N.ptr(ctx), N.ptr(out), N.ptr(P), B, HW, C, heads, ctx.shape[1], E, N.current_stream()))
  return out, P
