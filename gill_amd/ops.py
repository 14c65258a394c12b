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
         alpha: float = 1.0, act: str = "none", out_f32: bool = False, splitk: int = 0) -> torch.Tensor:
  """act(alpha * a @ w.T + bias + resid); a (M,K) bf16, w (N,K) bf16, bias (N) fp32, resid (M,N) bf16."""
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
