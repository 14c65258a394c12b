"""The helpers of gill/utils.py that the generation path touches: truncate_caption (gill/models.py:658, :759) and the image
pre-processing of PIL prompts (gill/utils.py:111-119, gill/models.py:606-613)."""
from __future__ import annotations

import os

import numpy as np
import torch


def truncate_caption(caption: str) -> str:
  """Truncate captions at periods and newlines.  (reference: gill/utils.py:32-40)"""
  caption = caption.strip('\n')
  trunc_index = caption.find('\n') + 1
  if trunc_index <= 0:
    trunc_index = caption.find('.') + 1
  if trunc_index > 0:
    caption = caption[:trunc_index]
  return caption


class ClipImageProcessor:
  """What `AutoFeatureExtractor.from_pretrained('openai/clip-vit-*')` (gill/utils.py:113) does to one PIL image, without the
  hub: RGB, bicubic resize of the shortest edge to `size`, centre crop to `crop_size`, scale to [0,1], normalise with the
  CLIP mean / std.  `__call__(img, return_tensors="pt").pixel_values` is (1,3,crop,crop) float32, like the HF object."""
  image_mean = (0.48145466, 0.4578275, 0.40821073)
  image_std = (0.26862954, 0.26130258, 0.27577711)

  def __init__(self, size: int = 224, crop_size: int = None):
    self.size = int(size)
    self.crop_size = int(crop_size if crop_size is not None else size)

  def __call__(self, img, return_tensors="pt"):
    from PIL import Image
    from types import SimpleNamespace
    img = img.convert('RGB')
    w, h = img.size
    short, long = (w, h) if w <= h else (h, w)
    new_short, new_long = self.size, int(self.size * long / short)
    nw, nh = (new_short, new_long) if w <= h else (new_long, new_short)
    img = img.resize((nw, nh), resample=Image.BICUBIC)
    c = self.crop_size
    top, left = (nh - c) // 2, (nw - c) // 2
    a = np.asarray(img, dtype=np.uint8)
    if top < 0 or left < 0:      # image smaller than the crop: zero-pad around the centre (as the HF centre crop does)
      ph, pw = max(c, nh), max(c, nw)
      pad = np.zeros((ph, pw, 3), dtype=np.uint8)
      pt, pl = (ph - nh) // 2, (pw - nw) // 2
      pad[pt:pt + nh, pl:pl + nw] = a
      a, nh, nw = pad, ph, pw
      top, left = (nh - c) // 2, (nw - c) // 2
    a = a[top:top + c, left:left + c].astype(np.float32) * np.float32(1.0 / 255.0)
    a = (a - np.asarray(self.image_mean, dtype=np.float32)) / np.asarray(self.image_std, dtype=np.float32)
    px = torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1)))[None]
    return SimpleNamespace(pixel_values=px)


def get_feature_extractor_for_model(model_name: str, image_size: int = 224, train: bool = True):
  """reference: gill/utils.py:111-114 (the CLIP pre-processing constants are the same for every openai/clip-vit-* model)."""
  print(f'Using the built-in CLIP image pre-processing for {model_name}.')
  return ClipImageProcessor(image_size, image_size)


def get_pixel_values_for_model(feature_extractor, img):
  """reference: gill/utils.py:117-119"""
  return feature_extractor(img.convert('RGB'), return_tensors="pt").pixel_values[0, ...]  # (3, H, W)


def get_image_from_url(url: str):
  """reference: gill/utils.py:24-29 (an http(s) URL through `requests`); additionally accepts a local file path, which is
  what an offline CC3M mirror provides.  Errors surface as OSError / UnidentifiedImageError like the reference's."""
  from io import BytesIO
  from PIL import Image
  if os.path.exists(url):
    img = Image.open(url)
  else:
    import requests
    response = requests.get(url)
    img = Image.open(BytesIO(response.content))
  img = img.resize((224, 224))
  img = img.convert('RGB')
  return img
