"""Oracle, image-prompt branch: the CLIP vision tower + GILL's visual projection, plain fp32 torch-CPU ops.

Reference call site: GILLModel.get_visual_embs (gill/models.py:129-152)
    outputs = self.visual_model(pixel_values) ; encoder_outputs = outputs.pooler_output
    visual_embs = self.visual_embeddings(encoder_outputs) ; reshape (B, n_visual_tokens, -1)
`visual_model` is transformers.CLIPVisionModel (gill/models.py:78-96); this file restates its forward (patch conv without
bias, class token, learned positions, pre_layrnorm, pre-LN encoder layers with quick_gelu MLPs, post_layernorm of token 0).
PINNED: tests/golden/gill_visual_tiny.npz holds the reference's own get_visual_embs output (oracle/gen_golden.py F5).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
"""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn.functional as F


def _ln(sd, p, x):
  return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"].float(), sd[p + ".bias"].float(), 1e-5)


def _lin(sd, p, x):
  return F.linear(x, sd[p + ".weight"].float(), sd[p + ".bias"].float())


def clip_pooler_output(sd: Dict[str, torch.Tensor], pixel_values: torch.Tensor, patch_size: int, num_heads: int) -> torch.Tensor:
  """pixel_values (B,3,S,S) -> pooler_output (B, D)."""
  vm = "vision_model"
  x = F.conv2d(pixel_values.float(), sd[f"{vm}.embeddings.patch_embedding.weight"].float(), stride=patch_size)
  B, D = x.shape[0], x.shape[1]
  x = x.flatten(2).transpose(1, 2)                                              # (B, P, D)
  cls = sd[f"{vm}.embeddings.class_embedding"].float().expand(B, 1, D)
  x = torch.cat([cls, x], dim=1) + sd[f"{vm}.embeddings.position_embedding.weight"].float()[None]
  x = _ln(sd, f"{vm}.pre_layrnorm", x)
  n_layers = 1 + max(int(k.split(".")[3]) for k in sd if k.startswith(f"{vm}.encoder.layers."))
  hd = D // num_heads
  for i in range(n_layers):
    p = f"{vm}.encoder.layers.{i}"
    h = _ln(sd, p + ".layer_norm1", x)
    q = _lin(sd, p + ".self_attn.q_proj", h) * hd ** -0.5
    k = _lin(sd, p + ".self_attn.k_proj", h)
    v = _lin(sd, p + ".self_attn.v_proj", h)
    T = x.shape[1]
    sh = lambda t: t.view(B, T, num_heads, hd).transpose(1, 2)                   # noqa: E731
    a = (sh(q) @ sh(k).transpose(-1, -2)).softmax(-1) @ sh(v)
    x = x + _lin(sd, p + ".self_attn.out_proj", a.transpose(1, 2).reshape(B, T, D))
    h = _lin(sd, p + ".mlp.fc1", _ln(sd, p + ".layer_norm2", x))
    h = h * torch.sigmoid(1.702 * h)                                            # quick_gelu
    x = x + _lin(sd, p + ".mlp.fc2", h)
  return _ln(sd, f"{vm}.post_layernorm", x[:, 0])


def visual_embs(clip_sd, proj_w, proj_b, pixel_values, patch_size, num_heads, n_visual_tokens):
  """get_visual_embs(mode='captioning'): (B, n_visual_tokens, lm_hidden)."""
  pooled = clip_pooler_output(clip_sd, pixel_values, patch_size, num_heads)
  e = F.linear(pooled, proj_w.float(), proj_b.float())
  return e.reshape(pixel_values.shape[0], n_visual_tokens, -1)
