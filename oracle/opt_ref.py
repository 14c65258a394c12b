"""Oracle, stage 1: the frozen OPT decoder forward, restated in plain fp32 torch-CPU ops.

Follows what the reference executes at gill/models.py:363-365 / :465:
    output = self.lm(inputs_embeds=input_embs, output_hidden_states=True)      # no attention_mask
    output.hidden_states[-1]                                                   # post final LayerNorm
whose arithmetic lives in the pinned dependency transformers==4.30.2 (requirements.txt:58),
modeling_opt.OPTDecoder / OPTDecoderLayer / OPTAttention (do_layer_norm_before=True, ReLU,
learned positions with offset 2, biases everywhere, tied lm_head).  tests/test_oracle_golden.py pins this
file against outputs of the reference's GILLModel running the installed transformers OPT.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
"""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn.functional as F


def opt_hidden_states(sd: Dict[str, torch.Tensor], num_layers: int, num_heads: int, inputs_embeds: torch.Tensor,
                      apply_final_ln: bool = True) -> torch.Tensor:
  """inputs_embeds (B,T,D) fp32 -> hidden_states[-1] (B,T,D) fp32."""
  x = inputs_embeds.float()
  B, T, D = x.shape
  hd = D // num_heads
  pre = "model.decoder."
  # OPTLearnedPositionalEmbedding: positions = cumsum(mask) - 1 + offset(2); full mask -> arange(T) + 2
  pos = sd[pre + "embed_positions.weight"].float()[2:2 + T]
  h = x + pos[None]
  causal = torch.full((T, T), float("-inf")).triu(1)
  for i in range(num_layers):
    p = f"{pre}layers.{i}."
    r = h
    y = F.layer_norm(h, (D,), sd[p + "self_attn_layer_norm.weight"].float(), sd[p + "self_attn_layer_norm.bias"].float(), 1e-5)
    q = F.linear(y, sd[p + "self_attn.q_proj.weight"].float(), sd[p + "self_attn.q_proj.bias"].float()) * hd ** -0.5
    k = F.linear(y, sd[p + "self_attn.k_proj.weight"].float(), sd[p + "self_attn.k_proj.bias"].float())
    v = F.linear(y, sd[p + "self_attn.v_proj.weight"].float(), sd[p + "self_attn.v_proj.bias"].float())
    q = q.view(B, T, num_heads, hd).transpose(1, 2)
    k = k.view(B, T, num_heads, hd).transpose(1, 2)
    v = v.view(B, T, num_heads, hd).transpose(1, 2)
    a = (q @ k.transpose(-1, -2) + causal).softmax(-1) @ v
    a = a.transpose(1, 2).reshape(B, T, D)
    h = r + F.linear(a, sd[p + "self_attn.out_proj.weight"].float(), sd[p + "self_attn.out_proj.bias"].float())
    r = h
    y = F.layer_norm(h, (D,), sd[p + "final_layer_norm.weight"].float(), sd[p + "final_layer_norm.bias"].float(), 1e-5)
    y = F.relu(F.linear(y, sd[p + "fc1.weight"].float(), sd[p + "fc1.bias"].float()))
    h = r + F.linear(y, sd[p + "fc2.weight"].float(), sd[p + "fc2.bias"].float())
  if apply_final_ln:
    h = F.layer_norm(h, (D,), sd[pre + "final_layer_norm.weight"].float(), sd[pre + "final_layer_norm.bias"].float(), 1e-5)
  return h


def opt_embed(sd: Dict[str, torch.Tensor], ids: torch.Tensor) -> torch.Tensor:
  """input_embeddings(ids) (gill/models.py:180, :620)."""
  return F.embedding(ids, sd["model.decoder.embed_tokens.weight"].float())


def opt_logits(sd: Dict[str, torch.Tensor], hidden: torch.Tensor) -> torch.Tensor:
  """lm_head(hidden_states[-1]); lm_head is tied to embed_tokens unless given."""
  w = sd.get("lm_head.weight", sd["model.decoder.embed_tokens.weight"]).float()
  return F.linear(hidden, w)
