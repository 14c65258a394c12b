"""Oracle, stage 3b: the Stable Diffusion VAE decoder + image post-processing, plain fp32 torch-CPU ops.

PARITY UNPINNED.  Reference call site: StableDiffusionPipeline.decode_latents (gill/custom_sd.py:385-392)
    latents = 1 / 0.18215 * latents ; image = self.vae.decode(latents).sample ; image = (image / 2 + 0.5).clamp(0, 1)
and numpy_to_pil's uint8 conversion.  The arithmetic is diffusers==0.17.1 AutoencoderKL (requirements.txt:9), absent from
the reference tree and from this image; this file restates the published decoder for the SD-1.5 `vae/config.json`:
post_quant_conv 1x1; conv_in 3x3 (4 -> 512); mid: ResnetBlock2D, single-head attention (GroupNorm(32, 1e-6), q/k/v/out
Linear with bias, scale C^-0.5, residual), ResnetBlock2D; 4 UpDecoderBlock2D over channels [512, 512, 256, 128], three
resnets each (no time embedding, eps 1e-6), nearest-2x Upsample2D + conv after the first three; GroupNorm + SiLU; conv_out.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
"""
from __future__ import annotations

from typing import Dict, Sequence

import torch
import torch.nn.functional as F


def _gn(sd, p, x, groups):
  return F.group_norm(x, groups, sd[p + ".weight"].float(), sd[p + ".bias"].float(), 1e-6)


def _conv(sd, p, x, padding=1):
  return F.conv2d(x, sd[p + ".weight"].float(), sd[p + ".bias"].float(), padding=padding)


def _resnet(sd, p, x, groups):
  h = _conv(sd, p + ".conv1", F.silu(_gn(sd, p + ".norm1", x, groups)))
  h = _conv(sd, p + ".conv2", F.silu(_gn(sd, p + ".norm2", h, groups)))
  if (p + ".conv_shortcut.weight") in sd:
    x = _conv(sd, p + ".conv_shortcut", x, padding=0)
  return x + h


def _attn(sd, p, x, groups):
  B, C, H, W = x.shape
  h = _gn(sd, p + ".group_norm", x, groups).permute(0, 2, 3, 1).reshape(B, H * W, C)
  lin = lambda n, t: F.linear(t, sd[f"{p}.{n}.weight"].float(), sd[f"{p}.{n}.bias"].float())  # noqa: E731
  q, k, v = lin("to_q", h), lin("to_k", h), lin("to_v", h)
  a = ((q @ k.transpose(-1, -2)) * C ** -0.5).softmax(-1) @ v
  o = lin("to_out.0", a).reshape(B, H, W, C).permute(0, 3, 1, 2)
  return x + o


def vae_decode(sd: Dict[str, torch.Tensor], latents: torch.Tensor, block_out_channels: Sequence[int] = (128, 256, 512, 512),
               groups: int = 32, scaling_factor: float = 0.18215) -> torch.Tensor:
  """latents (B,4,L,L) -> vae.decode(latents / scaling_factor).sample, (B,3,8L,8L) fp32."""
  z = latents.float() / scaling_factor
  z = _conv(sd, "post_quant_conv", z, padding=0)
  x = _conv(sd, "decoder.conv_in", z)
  x = _resnet(sd, "decoder.mid_block.resnets.0", x, groups)
  x = _attn(sd, "decoder.mid_block.attentions.0", x, groups)
  x = _resnet(sd, "decoder.mid_block.resnets.1", x, groups)
  for i in range(4):
    for j in range(3):
      x = _resnet(sd, f"decoder.up_blocks.{i}.resnets.{j}", x, groups)
    if i < 3:
      x = F.interpolate(x, scale_factor=2.0, mode="nearest")
      x = _conv(sd, f"decoder.up_blocks.{i}.upsamplers.0.conv", x)
  x = F.silu(_gn(sd, "decoder.conv_norm_out", x, groups))
  return _conv(sd, "decoder.conv_out", x)


def to_uint8(image: torch.Tensor) -> torch.Tensor:
  """(image / 2 + 0.5).clamp(0, 1) -> (B,H,W,3) uint8 as numpy_to_pil does: (x * 255).round()."""
  x = (image / 2 + 0.5).clamp(0, 1).permute(0, 2, 3, 1)
  return (x * 255).round().to(torch.uint8)
