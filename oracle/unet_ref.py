"""Oracle, stage 3a: the Stable Diffusion 1.5 UNet forward, restated in plain fp32 torch-CPU ops.

PARITY UNPINNED.  The reference calls  self.unet(latent_model_input, t, encoder_hidden_states=prompt_embeds).sample
(gill/custom_sd.py:633-638; at inference through the stock pipeline, gill/models.py:730-731); the arithmetic
is diffusers==0.17.1 UNet2DConditionModel (requirements.txt:9), a pinned dependency that is neither vendored
in /root/reference nor installed here, and the reference has no test/golden vector at this boundary.  This
file restates the published algorithm of that model for the SD-1.5 `unet/config.json`:

  Timesteps(320, flip_sin_to_cos=True, freq_shift=0) -> TimestepEmbedding(320->1280, SiLU, 1280->1280)
  conv_in 3x3; down: 3x CrossAttnDownBlock2D + DownBlock2D (2 ResnetBlock2D [+ Transformer2DModel] each,
  Downsample2D conv stride 2 pad 1); mid: resnet, transformer, resnet; up: UpBlock2D + 3x CrossAttnUpBlock2D
  (3 resnets over cat([h, skip]) each, Upsample2D nearest-2x + conv); GroupNorm(32, eps 1e-5)+SiLU; conv_out.
  ResnetBlock2D: GN+SiLU, conv1, + time_emb_proj(SiLU(temb)), GN+SiLU, conv2, + (1x1 shortcut if cin!=cout).
  Transformer2DModel (conv projections): GroupNorm(32, eps 1e-6), proj_in 1x1, BasicTransformerBlock
  (LN -> self-attn, LN -> cross-attn over the 77x768 context, LN -> GEGLU FF), proj_out 1x1, + residual.
  Attention: 8 heads ("attention_head_dim": 8 is the head COUNT), q/k/v without bias, to_out with bias,
  scale head_dim^-0.5.  GEGLU: h, gate = proj(x).chunk(2, -1); h * gelu(gate) (erf form).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
"""
from __future__ import annotations

import math
from typing import Dict, Sequence

import torch
import torch.nn.functional as F


def timestep_embedding(t: torch.Tensor, dim: int) -> torch.Tensor:
  half = dim // 2
  exponent = -math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half
  emb = t.float()[:, None] * torch.exp(exponent)[None, :]
  return torch.cat([torch.cos(emb), torch.sin(emb)], dim=-1)   # flip_sin_to_cos=True


def _gn(sd, p, x, groups, eps):
  return F.group_norm(x, groups, sd[p + ".weight"].float(), sd[p + ".bias"].float(), eps)


def _conv(sd, p, x, stride=1, padding=1):
  return F.conv2d(x, sd[p + ".weight"].float(), sd[p + ".bias"].float(), stride=stride, padding=padding)


def _lin(sd, p, x, bias=True):
  return F.linear(x, sd[p + ".weight"].float(), sd[p + ".bias"].float() if bias else None)


def _resnet(sd, p, x, temb, groups):
  h = _conv(sd, p + ".conv1", F.silu(_gn(sd, p + ".norm1", x, groups, 1e-5)))
  h = h + _lin(sd, p + ".time_emb_proj", F.silu(temb))[:, :, None, None]
  h = _conv(sd, p + ".conv2", F.silu(_gn(sd, p + ".norm2", h, groups, 1e-5)))
  if (p + ".conv_shortcut.weight") in sd:
    x = _conv(sd, p + ".conv_shortcut", x, padding=0)
  return x + h


def _attn(sd, p, x, ctx, heads):
  q, k, v = _lin(sd, p + ".to_q", x, False), _lin(sd, p + ".to_k", ctx, False), _lin(sd, p + ".to_v", ctx, False)
  B, N, C = q.shape
  M = k.shape[1]
  d = C // heads
  q = q.view(B, N, heads, d).transpose(1, 2)
  k = k.view(B, M, heads, d).transpose(1, 2)
  v = v.view(B, M, heads, d).transpose(1, 2)
  a = ((q @ k.transpose(-1, -2)) * d ** -0.5).softmax(-1) @ v
  return _lin(sd, p + ".to_out.0", a.transpose(1, 2).reshape(B, N, C))


def _ln(sd, p, x):
  return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"].float(), sd[p + ".bias"].float(), 1e-5)


def _transformer(sd, p, x, ctx, heads, groups):
  B, C, H, W = x.shape
  res = x
  h = _conv(sd, p + ".proj_in", _gn(sd, p + ".norm", x, groups, 1e-6), padding=0)
  h = h.permute(0, 2, 3, 1).reshape(B, H * W, C)
  b = p + ".transformer_blocks.0"
  h = h + _attn(sd, b + ".attn1", _ln(sd, b + ".norm1", h), _ln(sd, b + ".norm1", h), heads)
  h = h + _attn(sd, b + ".attn2", _ln(sd, b + ".norm2", h), ctx, heads)
  y = _lin(sd, b + ".ff.net.0.proj", _ln(sd, b + ".norm3", h))
  val, gate = y.chunk(2, dim=-1)
  h = h + _lin(sd, b + ".ff.net.2", val * F.gelu(gate))
  h = h.reshape(B, H, W, C).permute(0, 3, 1, 2)
  return _conv(sd, p + ".proj_out", h, padding=0) + res


def _h(heads, level):
  """heads: one count for every level (SD-1.x) or one per resolution level (SD-2.x: 5, 10, 20, 20)."""
  return heads if isinstance(heads, int) else heads[level]


def unet_forward(sd: Dict[str, torch.Tensor], sample: torch.Tensor, timesteps: torch.Tensor, ctx: torch.Tensor,
                 block_out_channels: Sequence[int] = (320, 640, 1280, 1280), heads=8, groups: int = 32) -> torch.Tensor:
  """sample (B,4,L,L), timesteps (B,), ctx (B,77,ctx_dim) -> predicted noise (B,4,L,L); all fp32."""
  ch = block_out_channels
  x = sample.float()
  ctx = ctx.float()
  temb = timestep_embedding(timesteps.expand(x.shape[0]) if timesteps.dim() == 0 else timesteps, ch[0])
  temb = _lin(sd, "time_embedding.linear_2", F.silu(_lin(sd, "time_embedding.linear_1", temb)))
  x = _conv(sd, "conv_in", x)
  skips = [x]
  for i in range(4):
    for j in range(2):
      x = _resnet(sd, f"down_blocks.{i}.resnets.{j}", x, temb, groups)
      if i < 3:
        x = _transformer(sd, f"down_blocks.{i}.attentions.{j}", x, ctx, _h(heads, i), groups)
      skips.append(x)
    if i < 3:
      x = _conv(sd, f"down_blocks.{i}.downsamplers.0.conv", x, stride=2, padding=1)
      skips.append(x)
  x = _resnet(sd, "mid_block.resnets.0", x, temb, groups)
  x = _transformer(sd, "mid_block.attentions.0", x, ctx, _h(heads, 3), groups)
  x = _resnet(sd, "mid_block.resnets.1", x, temb, groups)
  for i in range(4):
    for j in range(3):
      x = torch.cat([x, skips.pop()], dim=1)
      x = _resnet(sd, f"up_blocks.{i}.resnets.{j}", x, temb, groups)
      if i > 0:
        x = _transformer(sd, f"up_blocks.{i}.attentions.{j}", x, ctx, _h(heads, 3 - i), groups)
    if i < 3:
      x = F.interpolate(x, scale_factor=2.0, mode="nearest")
      x = _conv(sd, f"up_blocks.{i}.upsamplers.0.conv", x)
  x = F.silu(_gn(sd, "conv_norm_out", x, groups, 1e-5))
  return _conv(sd, "conv_out", x)
