"""NSFW filter of the Stable Diffusion pipeline: what `self.run_safety_checker(image, device, dtype)` does at
gill/custom_sd.py:375-383 / :657 when the pipeline was loaded with its `safety_checker` component (the reference's
`StableDiffusionPipeline.from_pretrained("runwayml/stable-diffusion-v1-5")` loads it by default, gill/models.py:550-551).

The checker itself is `diffusers.pipelines.stable_diffusion.safety_checker.StableDiffusionSafetyChecker` ([DEP] diffusers==0.17.1,
absent from the reference tree and from this image — PARITY UNPINNED, restated from the published algorithm):
  pooled = CLIP ViT-L/14 vision tower's pooler_output of the pre-processed image      -> gill_clip_forward (csrc/clip.hip)
  e      = visual_projection(pooled), unit-normalised                                 -> gill_op_gemm
  cos    = e . concept_embeds^T (17 rows) and e . special_care_embeds^T (3 rows), rows unit-normalised   -> gill_op_gemm
  per image: adjustment = 0; every special-care concept whose (cos - threshold + adjustment) rounded to 3 decimals is > 0 sets
  adjustment = 0.01; the image is flagged when any concept has round(cos - threshold + adjustment, 3) > 0
  flagged images are replaced by black images.
State-dict keys (safety_checker/model.safetensors): vision_model.vision_model.*, visual_projection.weight (768, 1024),
concept_embeds (17, 768), special_care_embeds (3, 768), concept_embeds_weights (17), special_care_embeds_weights (3).
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Tuple

import numpy as np
import torch

from . import _native as N
from .synth import ClipConfig
from .utils import ClipImageProcessor


class GillSafetyChecker:
  def __init__(self, state: Dict[str, torch.Tensor], clip_cfg: ClipConfig, device, max_batch: int = 8):
    self.cfg = clip_cfg
    self.device = torch.device(device)
    self.max_batch = max_batch
    vis = {}
    for k, v in state.items():
      if k.startswith("vision_model."):
        kk = k[len("vision_model."):]
        vis[kk if kk.startswith("vision_model.") else "vision_model." + kk] = v
    cc = N.gill_clip_config(image_size=clip_cfg.image_size, patch_size=clip_cfg.patch_size, hidden_size=clip_cfg.hidden_size,
                            num_layers=clip_cfg.num_layers, num_heads=clip_cfg.num_heads,
                            intermediate_size=clip_cfg.intermediate_size, max_batch=max_batch)
    arr, keep = N.make_tensor_table(vis, self.device)
    h = C.c_void_p()
    with torch.cuda.device(self.device):
      N.check(N.lib().gill_clip_create(C.byref(h), C.byref(cc), arr, len(vis)))
    del keep
    self._h = h
    dev = self.device
    self.proj = state["visual_projection.weight"].to(dev, torch.bfloat16).contiguous()           # (P, hidden)
    # concept rows are unit-normalised once (cosine_distance normalises both sides); special-care rows first
    both = torch.cat([state["special_care_embeds"].float(), state["concept_embeds"].float()], 0)
    both = both / both.norm(dim=1, keepdim=True)
    self.n_special = state["special_care_embeds"].shape[0]
    pad = (-both.shape[0]) % 4                     # the GEMM epilogue writes 4 columns at a time
    self.concepts = torch.cat([both, torch.zeros(pad, both.shape[1])], 0).to(dev, torch.bfloat16).contiguous()
    self.n_concepts = both.shape[0]
    self.thresholds = torch.cat([state["special_care_embeds_weights"].float(), state["concept_embeds_weights"].float()]).cpu().numpy()
    self.feature_extractor = ClipImageProcessor(clip_cfg.image_size, clip_cfg.image_size)

  def __del__(self):
    try:
      if getattr(self, "_h", None):
        N.lib().gill_clip_destroy(self._h)
        self._h = None
    except Exception:
      pass

  def cosines(self, pixel_values: torch.Tensor) -> np.ndarray:
    """(B,3,S,S) pre-processed pixels -> (B, n_special + n_concepts) cosine similarities, special-care columns first."""
    from . import ops
    px = pixel_values.to(self.device, torch.float32).contiguous()
    B = px.shape[0]
    out = []
    for i in range(0, B, self.max_batch):
      b = min(self.max_batch, B - i)
      pooled = torch.empty((b, self.cfg.hidden_size), device=self.device, dtype=torch.float32)
      with torch.cuda.device(self.device):
        N.check(N.lib().gill_clip_forward(self._h, N.ptr(px[i:i + b]), b, N.ptr(pooled), N.current_stream()))
      e = ops.gemm(pooled.to(torch.bfloat16), self.proj, out_f32=True)
      e = e / e.norm(dim=1, keepdim=True)
      out.append(ops.gemm(e.to(torch.bfloat16), self.concepts, out_f32=True)[:, :self.n_concepts])
    return torch.cat(out, 0).float().cpu().numpy()

  @staticmethod
  def decide(cos: np.ndarray, thresholds: np.ndarray, n_special: int) -> List[bool]:
    """The reference checker's per-image decision on the cosine rows (host control logic, like the original)."""
    flags = []
    for row in cos:
      adjustment = 0.0
      for c in range(n_special):
        if round(float(row[c]) - float(thresholds[c]) + adjustment, 3) > 0:
          adjustment = 0.01
      bad = False
      for c in range(n_special, len(row)):
        if round(float(row[c]) - float(thresholds[c]) + adjustment, 3) > 0:
          bad = True
      flags.append(bad)
    return flags

  def __call__(self, images: np.ndarray, pil_images) -> Tuple[np.ndarray, List[bool]]:
    """images (B,H,W,3) float in [0,1] (decode_latents' output), pil_images the same as PIL -> (images with flagged ones
    blacked out, has_nsfw_concept)."""
    px = torch.cat([self.feature_extractor(im, return_tensors="pt").pixel_values for im in pil_images], 0)
    flags = self.decide(self.cosines(px), self.thresholds, self.n_special)
    images = np.array(images, copy=True)
    for i, bad in enumerate(flags):
      if bad:
        images[i] = np.zeros(images[i].shape, dtype=images.dtype)
    if any(flags):
      print("Potential NSFW content was detected in one or more images. A black image will be returned instead."
            " Try again with a different prompt and/or seed.")
    return images, flags
