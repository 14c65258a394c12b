"""Oracle, stage 2: GILLMapper = gill.layers.TextFcLayer(mode='gill_mapper').forward (gill/layers.py:28-53),
restated in plain fp32 torch-CPU ops (no nn.Transformer / nn.MultiheadAttention modules, so the fused
inference fast paths of torch cannot hide a semantic difference):

    x = x + input_embs                                                         layers.py:31-32
    x = self.fc(x)                                                             layers.py:42
    x = self.tfm(x, self.query_embs.repeat(x.shape[0], 1, 1))                  layers.py:43
        nn.Transformer(batch_first, norm_first, d_model=512, nhead=4, 4 enc / 4 dec layers,
                       dim_feedforward=2048, dropout=0, ReLU), no masks        layers.py:19-21
    outputs = self.model(x)                                                    layers.py:44

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Pinned by tests/golden/mapper_*.npz.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.nn.functional as F


def _mha(sd, p, xq, xkv, nhead):
  E = xq.shape[-1]
  w, b = sd[p + ".in_proj_weight"].float(), sd[p + ".in_proj_bias"].float()
  q = F.linear(xq, w[:E], b[:E])
  k = F.linear(xkv, w[E:2 * E], b[E:2 * E])
  v = F.linear(xkv, w[2 * E:], b[2 * E:])
  B, Tq, _ = q.shape
  Tk = k.shape[1]
  hd = E // nhead
  q = q.view(B, Tq, nhead, hd).transpose(1, 2) * hd ** -0.5
  k = k.view(B, Tk, nhead, hd).transpose(1, 2)
  v = v.view(B, Tk, nhead, hd).transpose(1, 2)
  a = (q @ k.transpose(-1, -2)).softmax(-1) @ v
  a = a.transpose(1, 2).reshape(B, Tq, E)
  return F.linear(a, sd[p + ".out_proj.weight"].float(), sd[p + ".out_proj.bias"].float())


def _ln(sd, p, x):
  return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"].float(), sd[p + ".bias"].float(), 1e-5)


def _ff(sd, p, x):
  return F.linear(F.relu(F.linear(x, sd[p + ".linear1.weight"].float(), sd[p + ".linear1.bias"].float())),
                  sd[p + ".linear2.weight"].float(), sd[p + ".linear2.bias"].float())


def mapper_forward(sd: Dict[str, torch.Tensor], x: torch.Tensor, input_embs: Optional[torch.Tensor], nhead: int = 4,
                   num_enc: int = 4, num_dec: int = 4) -> torch.Tensor:
  """x (B,8,in_dim), input_embs (B|1,8,in_dim) -> (B,77,out_dim), fp32."""
  x = x.float()
  if input_embs is not None:
    x = x + input_embs.float()
  h = F.linear(x, sd["fc.weight"].float(), sd["fc.bias"].float())
  for i in range(num_enc):
    p = f"tfm.encoder.layers.{i}"
    y = _ln(sd, p + ".norm1", h)
    h = h + _mha(sd, p + ".self_attn", y, y, nhead)
    h = h + _ff(sd, p, _ln(sd, p + ".norm2", h))
  mem = _ln(sd, "tfm.encoder.norm", h)
  t = sd["query_embs"].float().repeat(x.shape[0], 1, 1)
  for i in range(num_dec):
    p = f"tfm.decoder.layers.{i}"
    y = _ln(sd, p + ".norm1", t)
    t = t + _mha(sd, p + ".self_attn", y, y, nhead)
    t = t + _mha(sd, p + ".multihead_attn", _ln(sd, p + ".norm2", t), mem, nhead)
    t = t + _ff(sd, p, _ln(sd, p + ".norm3", t))
  t = _ln(sd, "tfm.decoder.norm", t)
  return F.linear(t, sd["model.weight"].float(), sd["model.bias"].float())
