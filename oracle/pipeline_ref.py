"""Oracle: the glue of the hot path — the batched one-pass [IMG] extraction of
GILLModel.forward(mode='generation') (gill/models.py:180-183, 363-365, 384-387, 418) and the
classifier-free-guidance denoise loop (gill/custom_sd.py:607-651), on top of the stage oracles.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence

import torch

from . import mapper_ref, opt_ref, scheduler_ref, unet_ref


def img_hidden_and_embeds(opt_sd: Dict[str, torch.Tensor], num_layers: int, num_heads: int, ids: torch.Tensor,
                          last_idx: torch.Tensor, num_tokens: int = 8):
  """ids (B,T) right-padded, last_idx (B,) = caption_len - 1  ->  (raw (B,8,D), embs (B,8,D)).
  models.py:180 input_embs = input_embeddings(labels); :363 lm(inputs_embeds=input_embs) WITHOUT attention mask;
  :384 hidden_states[-1][i, idx-7:idx+1]; :385 input_embs[i, idx-7:idx+1]."""
  emb = opt_ref.opt_embed(opt_sd, ids)
  hid = opt_ref.opt_hidden_states(opt_sd, num_layers, num_heads, emb)
  raw = torch.stack([hid[i, int(last_idx[i]) - num_tokens + 1:int(last_idx[i]) + 1] for i in range(ids.shape[0])], 0)
  e = torch.stack([emb[i, int(last_idx[i]) - num_tokens + 1:int(last_idx[i]) + 1] for i in range(ids.shape[0])], 0)
  return raw, e


def sd_embedding(opt_sd, mapper_sd, num_layers: int, num_heads: int, ids: torch.Tensor, last_idx: torch.Tensor,
                 round_bf16: bool = False) -> torch.Tensor:
  """prompt ids -> (B,77,768) SD conditioning = last_embedding of GILLModel.forward(mode='generation') (models.py:387,418)."""
  raw, e = img_hidden_and_embeds(opt_sd, num_layers, num_heads, ids, last_idx)
  if round_bf16:  # the reference model is .bfloat16(): the mapper sees bf16 hidden states (models.py:876)
    raw, e = raw.bfloat16().float(), e.bfloat16().float()
  return mapper_ref.mapper_forward(mapper_sd, raw, e)


def denoise(unet_sd: Dict[str, torch.Tensor], cond: torch.Tensor, uncond: Optional[torch.Tensor], latents: torch.Tensor,
            num_inference_steps: int = 50, guidance_scale: float = 7.5,
            block_out_channels: Sequence[int] = (320, 640, 1280, 1280), heads=8, groups: int = 32,
            return_eps: bool = False, prediction_type: str = "epsilon"):
  """custom_sd.py:588-651 (without VAE decode): cond (B,77,768), uncond (1,77,768) or per-sample (B,77,768), latents (B,4,L,L)
  fp32.  Pinned to the reference's own driver lines by tests/golden/sd_driver_tiny.npz (oracle/gen_golden.py F8)."""
  B = cond.shape[0]
  do_cfg = guidance_scale > 1.0
  if do_cfg and uncond.shape[0] != B:
    uncond = uncond.expand(B, -1, -1)
  ctx = torch.cat([uncond, cond], 0) if do_cfg else cond                        # custom_sd.py:371
  sched = scheduler_ref.PNDMSchedulerRef(prediction_type=prediction_type)
  sched.set_timesteps(num_inference_steps)                                      # :607
  lat = latents.float() * sched.init_noise_sigma                                # :472
  eps_trace = []
  for t in sched.timesteps:                                                     # :628
    inp = torch.cat([lat] * 2) if do_cfg else lat                               # :630
    inp = sched.scale_model_input(inp, t)                                       # :631
    eps = unet_ref.unet_forward(unet_sd, inp, torch.full((inp.shape[0],), float(t)), ctx, block_out_channels, heads, groups)
    if do_cfg:
      eu, ec = eps.chunk(2)
      eps = eu + guidance_scale * (ec - eu)                                     # :641-643
    if return_eps:
      eps_trace.append(eps)
    lat = sched.step(eps, t, lat)                                               # :646
  return (lat, eps_trace) if return_eps else lat
