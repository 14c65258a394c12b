"""Oracle: StableDiffusionSafetyChecker.forward ([DEP] diffusers==0.17.1 — absent here: PARITY UNPINNED), the object behind
`self.run_safety_checker` (gill/custom_sd.py:375-383).  Plain fp32 torch-CPU ops on top of oracle/clip_ref.py.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import torch

from . import clip_ref


def cosine_distance(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
  a = torch.nn.functional.normalize(a)
  b = torch.nn.functional.normalize(b)
  return a @ b.t()


def safety_check(sd: Dict[str, torch.Tensor], pixel_values: torch.Tensor, num_heads: int, patch: int
                 ) -> Tuple[List[bool], torch.Tensor, torch.Tensor]:
  """-> (has_nsfw_concept per image, special-care cosines (B,3), concept cosines (B,17))."""
  vis = {}
  for k, v in sd.items():
    if k.startswith("vision_model."):
      kk = k[len("vision_model."):]
      vis[kk if kk.startswith("vision_model.") else "vision_model." + kk] = v
  pooled = clip_ref.clip_pooler_output(vis, pixel_values.float(), patch, num_heads)
  emb = pooled @ sd["visual_projection.weight"].float().t()
  special = cosine_distance(emb, sd["special_care_embeds"].float())
  cos = cosine_distance(emb, sd["concept_embeds"].float())
  flags = []
  for i in range(emb.shape[0]):
    adjustment = 0.0
    for c in range(special.shape[1]):
      if round(float(special[i, c]) - float(sd["special_care_embeds_weights"][c]) + adjustment, 3) > 0:
        adjustment = 0.01
    bad = any(round(float(cos[i, c]) - float(sd["concept_embeds_weights"][c]) + adjustment, 3) > 0 for c in range(cos.shape[1]))
    flags.append(bad)
  return flags, special, cos
