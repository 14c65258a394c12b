"""Stable Diffusion side of the hot path: the object GILL holds as `self.sd_pipe`.

The reference builds `StableDiffusionPipeline.from_pretrained("runwayml/stable-diffusion-v1-5", fp16).to("cuda")`
(gill/models.py:550-551) and calls it as
    self.sd_pipe(prompt_embeds=gen_emb[i:i+8], generator=generator, guidance_scale=..., num_inference_steps=...).images
(gill/models.py:730-731); the loop it runs is restated in-tree at gill/custom_sd.py:567-651.
`GillSDPipeline` keeps that call signature (prompt_embeds / negative_prompt_embeds / latents / generator /
guidance_scale / num_inference_steps / output_type) and runs the UNet + PNDM loop in libgill_amd
(gill_sd_denoise: csrc/unet.hip).

`output_type` defaults to "pil" like the reference's (gill/custom_sd.py:491) when the handle holds VAE weights (a UNet-only
handle defaults to "latent"); "latent" (what `GILL.generate_images` passes: the hot path hands latents to the all-gather) returns
the final latents (B,4,64,64) fp32 in `.images`; "pil" / "np" / "pt" run the VAE decoder (gill_vae_decode: csrc/vae.hip, custom_sd.py:385-392, :654-661) and
need the handle to have been built with VAE weights.
"""
from __future__ import annotations

import ctypes as C
import json
import os
from dataclasses import dataclass
from typing import Dict, List, Optional, Union

import torch

from . import _native as N
from .synth import UNetConfig, VAEConfig


@dataclass
class PipelineOutput:
  images: object                      # latents tensor (B,4,L,L) when output_type == "latent"
  nsfw_content_detected: Optional[List[bool]] = None


class GillSDPipeline:
  def __init__(self, unet_state: Dict[str, torch.Tensor], cfg: UNetConfig, uncond_embeds: torch.Tensor,
               device: Union[str, torch.device] = "cuda", max_batch: int = 16,
               vae_state: Optional[Dict[str, torch.Tensor]] = None, vae_cfg: Optional[VAEConfig] = None):
    self.cfg = cfg
    self.device = torch.device(device if str(device) != "cuda" else "cuda:0")
    if self.device.type != "cuda":
      raise N.GillNativeError("GillSDPipeline runs only on an MI355X through libgill_amd")
    self.max_batch = max_batch
    #  the captured denoise loop wants DEBUG_CLR_GRAPH_PACKET_CAPTURE=0, which ROCm reads when HIP initialises: entry points call
    # gill_amd.configure_hip_runtime() first; a caller that did not gets ONE RuntimeWarning here instead of a silent 1.2 % (README "Runtime setting")
    from . import configure_hip_runtime
    configure_hip_runtime(warn=True)
    self.uncond_embeds = uncond_embeds.to(self.device, torch.bfloat16).reshape(1, cfg.ctx_len, cfg.cross_attention_dim).contiguous()
    ccfg = N.gill_unet_config(in_channels=cfg.in_channels, out_channels=cfg.out_channels,
                              layers_per_block=cfg.layers_per_block, cross_attention_dim=cfg.cross_attention_dim,
                              num_heads=cfg.num_heads, norm_num_groups=cfg.norm_num_groups,
                              v_prediction=int(cfg.prediction_type == "v_prediction"), fp8_convs=int(cfg.fp8_convs),
                              sample_size=cfg.sample_size, ctx_len=cfg.ctx_len, max_batch=max_batch)
    for i in range(4):
      ccfg.block_out_channels[i] = cfg.block_out_channels[i]
      ccfg.heads_per_level[i] = cfg.heads_per_level[i] if cfg.heads_per_level else 0
    arr, keep = N.make_tensor_table(unet_state, self.device)
    h = C.c_void_p()
    with torch.cuda.device(self.device):
      N.check(N.lib().gill_unet_create(C.byref(h), C.byref(ccfg), arr, len(unet_state)))
    del keep
    self._h = h
    self._vae = None
    self.vae_cfg = None
    self.safety_checker = None          # GillSafetyChecker (custom_sd.py:375-383), loaded by from_pretrained when its files exist
    if vae_state is not None:
      self.load_vae(vae_state, vae_cfg or VAEConfig(latent_size=cfg.sample_size))

  def load_vae(self, vae_state: Dict[str, torch.Tensor], vae_cfg: VAEConfig) -> None:
    """AutoencoderKL decoder half (state-dict keys post_quant_conv.* / decoder.*) -> gill_vae handle."""
    v = N.gill_vae_config(latent_channels=vae_cfg.latent_channels, out_channels=vae_cfg.out_channels,
                          layers_per_block=vae_cfg.layers_per_block, norm_num_groups=vae_cfg.norm_num_groups,
                          latent_size=vae_cfg.latent_size, scaling_factor=vae_cfg.scaling_factor,
                          max_batch=max(1, self.max_batch // 2))
    for i in range(4):
      v.block_out_channels[i] = vae_cfg.block_out_channels[i]
    dec = {k: t for k, t in vae_state.items() if k.startswith("decoder.") or k.startswith("post_quant_conv.")}
    arr, keep = N.make_tensor_table(dec, self.device)
    h = C.c_void_p()
    with torch.cuda.device(self.device):
      N.check(N.lib().gill_vae_create(C.byref(h), C.byref(v), arr, len(dec)))
    del keep
    self._vae, self.vae_cfg = h, vae_cfg

  def decode_latents(self, latents: torch.Tensor, as_uint8: bool = True, both: bool = False):
    """custom_sd.py:385-392 (+ numpy_to_pil's uint8 step): latents (B,4,L,L) -> (B,8L,8L,3) uint8, or with
    as_uint8=False the raw decoder output vae.decode(latents / 0.18215).sample (B,3,8L,8L) fp32; both=True returns
    (fp32, uint8) of the same decode."""
    if self._vae is None:
      raise N.GillNativeError("this pipeline was built without VAE weights (pass vae_state= / use from_pretrained)")
    lat = latents.to(self.device, torch.float32).contiguous()
    B, side = lat.shape[0], 8 * self.vae_cfg.latent_size
    ch = self.vae_cfg.out_channels
    f32 = torch.empty((B, ch, side, side), device=self.device, dtype=torch.float32) if (both or not as_uint8) else None
    u8 = torch.empty((B, side, side, ch), device=self.device, dtype=torch.uint8) if (both or as_uint8) else None
    cap = max(1, self.max_batch // 2)
    with torch.cuda.device(self.device):
      for i in range(0, B, cap):
        b = min(cap, B - i)
        N.check(N.lib().gill_vae_decode(self._vae, N.ptr(lat[i:i + b]), b, None if f32 is None else N.ptr(f32[i:i + b]),
                                        None if u8 is None else N.ptr(u8[i:i + b]), N.current_stream()))
    return (f32, u8) if both else (u8 if as_uint8 else f32)

  # ---- construction from a local diffusers directory (no diffusers import: safetensors + json only)
  @classmethod
  def from_pretrained(cls, model_dir: str, uncond_embeds: Optional[torch.Tensor] = None, device="cuda", max_batch: int = 16,
                      **_ignored):
    from safetensors.torch import load_file
    with open(os.path.join(model_dir, "unet", "config.json")) as f:
      c = json.load(f)
    ahd = c["attention_head_dim"]     # diffusers quirk: this field holds the head COUNT(s)
    heads = ahd if isinstance(ahd, int) else ahd[0]
    pred = "epsilon"
    spath = os.path.join(model_dir, "scheduler", "scheduler_config.json")
    if os.path.exists(spath):
      with open(spath) as f:
        pred = json.load(f).get("prediction_type", "epsilon")
    cfg = UNetConfig(in_channels=c["in_channels"], out_channels=c["out_channels"],
                     block_out_channels=tuple(c["block_out_channels"]), layers_per_block=c["layers_per_block"],
                     cross_attention_dim=c["cross_attention_dim"], num_heads=heads,
                     norm_num_groups=c["norm_num_groups"], sample_size=c["sample_size"],
                     heads_per_level=None if isinstance(ahd, int) else tuple(ahd), prediction_type=pred)
    wpath = os.path.join(model_dir, "unet", "diffusion_pytorch_model.safetensors")
    sd = load_file(wpath)
    upath = os.path.join(model_dir, "uncond_embeds.safetensors")
    if uncond_embeds is None and os.path.exists(upath):
      # offline cache of CLIP-text("") (1,77,ctx_dim): what _encode_prompt computes for the empty negative prompt
      # (custom_sd.py:319-357), for machines without the text encoder / tokenizer files
      uncond_embeds = load_file(upath)["uncond_embeds"]
    if uncond_embeds is None:
      # CLIP-text("") — plumbing through transformers when its files are on disk
      from transformers import CLIPTextModel, CLIPTokenizer
      tok = CLIPTokenizer.from_pretrained(os.path.join(model_dir, "tokenizer"))
      enc = CLIPTextModel.from_pretrained(os.path.join(model_dir, "text_encoder"))
      with torch.no_grad():
        ids = tok([""], padding="max_length", max_length=tok.model_max_length, return_tensors="pt").input_ids
        uncond_embeds = enc(ids)[0]
    vae_sd, vae_cfg = None, None
    vdir = os.path.join(model_dir, "vae")
    if os.path.exists(os.path.join(vdir, "diffusion_pytorch_model.safetensors")):
      with open(os.path.join(vdir, "config.json")) as f:
        vc = json.load(f)
      vae_cfg = VAEConfig(latent_channels=vc["latent_channels"], out_channels=vc["out_channels"],
                          block_out_channels=tuple(vc["block_out_channels"]), layers_per_block=vc["layers_per_block"],
                          norm_num_groups=vc["norm_num_groups"], latent_size=cfg.sample_size)   # scaling 0.18215: custom_sd.py:387
      vae_sd = load_file(os.path.join(vdir, "diffusion_pytorch_model.safetensors"))
    pipe = cls(sd, cfg, uncond_embeds, device, max_batch, vae_state=vae_sd, vae_cfg=vae_cfg)
    sdir = os.path.join(model_dir, "safety_checker")
    if os.path.exists(os.path.join(sdir, "model.safetensors")):     # the reference's from_pretrained loads it by default
      from .safety import GillSafetyChecker
      from .synth import ClipConfig
      with open(os.path.join(sdir, "config.json")) as f:
        vc = json.load(f).get("vision_config", {})
      ccfg = ClipConfig(image_size=vc.get("image_size", 224), patch_size=vc.get("patch_size", 14), hidden_size=vc.get("hidden_size", 1024),
                        num_layers=vc.get("num_hidden_layers", 24), num_heads=vc.get("num_attention_heads", 16),
                        intermediate_size=vc.get("intermediate_size", 4096))
      pipe.safety_checker = GillSafetyChecker(load_file(os.path.join(sdir, "model.safetensors")), ccfg, pipe.device, max_batch=max(1, max_batch // 2))
    return pipe

  def to(self, device):   # the reference chains .to("cuda")
    assert torch.device(device).type == "cuda"
    return self

  def __del__(self):
    try:
      if getattr(self, "_h", None):
        N.lib().gill_unet_destroy(self._h)
        self._h = None
      if getattr(self, "_vae", None):
        N.lib().gill_vae_destroy(self._vae)
        self._vae = None
    except Exception:
      pass

  # ---- low level: one UNet forward (used by parity tests)
  def unet(self, sample: torch.Tensor, timesteps: torch.Tensor, encoder_hidden_states: torch.Tensor) -> torch.Tensor:
    Bx = sample.shape[0]
    sample = sample.to(self.device, torch.float32).contiguous()
    ctx = encoder_hidden_states.to(self.device, torch.bfloat16).contiguous()
    ts = timesteps.detach().float().cpu().reshape(-1)
    if ts.numel() == 1:
      ts = ts.expand(Bx).contiguous()
    tarr = (C.c_float * Bx)(*[float(v) for v in ts])
    out = torch.empty_like(sample)
    with torch.cuda.device(self.device):
      N.check(N.lib().gill_unet_forward(self._h, N.ptr(sample), tarr, N.ptr(ctx), Bx, N.ptr(out), N.current_stream()))
    return out

  def prepare_latents(self, batch_size: int, generator=None, latents: Optional[torch.Tensor] = None) -> torch.Tensor:
    """custom_sd.py:458-473.  Latents are drawn on the CPU generator (cuRAND Philox streams of the reference
    notebooks are not reproducible off-NVIDIA); init_noise_sigma == 1 for PNDM."""
    L = self.cfg.sample_size
    shape = (batch_size, self.cfg.in_channels, L, L)
    if isinstance(generator, list) and len(generator) != batch_size:
      raise ValueError(f"You have passed a list of generators of length {len(generator)}, but requested an effective batch"
                       f" size of {batch_size}. Make sure the batch size matches the length of the generators.")
    if latents is None:
      if isinstance(generator, list):
        latents = torch.cat([torch.randn((1,) + shape[1:], generator=g, device=g.device if g is not None else "cpu",
                                         dtype=torch.float32) for g in generator], 0)
      else:
        gdev = generator.device if generator is not None else torch.device("cpu")
        latents = torch.randn(shape, generator=generator, device=gdev, dtype=torch.float32)
    return latents.to(self.device, torch.float32).contiguous()

  @torch.no_grad()
  def __call__(self, prompt=None, height=None, width=None, num_inference_steps: int = 50, guidance_scale: float = 7.5,
               negative_prompt=None, num_images_per_prompt: int = 1, eta: float = 0.0, generator=None,
               latents: Optional[torch.Tensor] = None, prompt_embeds: Optional[torch.Tensor] = None,
               negative_prompt_embeds: Optional[torch.Tensor] = None, output_type: Optional[str] = None, return_dict: bool = True,
               **_ignored):
    if prompt is not None:
      raise ValueError("GillSDPipeline is driven by prompt_embeds (gill/models.py:730); text prompts need the CLIP text "
                       "encoder, which is outside this path")
    if prompt_embeds is None:
      raise ValueError("Provide either `prompt` or `prompt_embeds`. Cannot leave both `prompt` and `prompt_embeds` undefined.")
    if output_type is None:
      # the reference's default is "pil" (gill/custom_sd.py:491: `.images` is a list of PIL images); a handle built WITHOUT VAE
      # weights (UNet-only parity rigs) can only return latents
      output_type = "pil" if self._vae is not None else "latent"
    L = self.cfg.sample_size
    if (height is not None and height != L * 8) or (width is not None and width != L * 8):
      raise ValueError(f"this handle was built for {L * 8}x{L * 8} images")
    if output_type not in ("latent", "pil", "np", "pt"):
      raise ValueError(f"unknown output_type {output_type!r}")
    if output_type != "latent" and self._vae is None:
      raise N.GillNativeError("output_type != 'latent' needs the VAE decoder weights (vae_state= / from_pretrained)")
    cond = prompt_embeds.to(self.device, torch.bfloat16)
    if num_images_per_prompt != 1:
      cond = cond.repeat_interleave(num_images_per_prompt, dim=0)
    cond = cond.contiguous()
    B = cond.shape[0]
    if negative_prompt_embeds is None:
      uncond = self.uncond_embeds
    else:      # custom_sd.py:359-369: one negative embedding per prompt (or one for all), repeated per image
      uncond = negative_prompt_embeds.to(self.device, torch.bfloat16)
      if uncond.shape[0] != 1 and num_images_per_prompt != 1:
        uncond = uncond.repeat_interleave(num_images_per_prompt, dim=0)
      uncond = uncond.contiguous()
      if uncond.shape[0] not in (1, B) or tuple(uncond.shape[1:]) != tuple(cond.shape[1:]):
        raise ValueError(f"`negative_prompt_embeds` must have the shape of `prompt_embeds` (or batch 1): got "
                         f"{tuple(negative_prompt_embeds.shape)} for prompt_embeds {tuple(prompt_embeds.shape)}")
    lat0 = self.prepare_latents(B, generator, latents)
    out = torch.empty_like(lat0)
    with torch.cuda.device(self.device):
      N.check(N.lib().gill_sd_denoise(self._h, N.ptr(cond), N.ptr(uncond), int(uncond.shape[0]), N.ptr(lat0), B,
                                      int(num_inference_steps), float(guidance_scale), N.ptr(out), N.current_stream()))
    has_nsfw = None
    if output_type in ("pil", "np"):      # custom_sd.py:654-661: decode_latents -> run_safety_checker -> numpy_to_pil
      from PIL import Image
      u8 = self.decode_latents(out, as_uint8=True).cpu().numpy()
      n = N.lib().gill_coop_timeouts()      # (the copy above synchronised: the loop's give-up count is final — include/gill_amd.h "Exclusive-device contract")
      if n != 0:
        raise N.GillNativeError(f"{n} in-kernel GroupNorm finish(es) of the denoise loop timed out and NaN-poisoned the latents: the GPU is shared with "
                                "another stream or process that also runs waiting workgroups; give this pipeline the device, or set GILL_GEMM_COOP=0")
      if self.safety_checker is not None:
        img01, has_nsfw = self.safety_checker(u8.astype("float32") / 255.0, [Image.fromarray(im) for im in u8])
        u8 = (img01 * 255).round().astype("uint8")
      out = [Image.fromarray(im) for im in u8] if output_type == "pil" else u8.astype("float32") / 255.0
    elif output_type == "pt":
      out = self.decode_latents(out, as_uint8=False)
    if not return_dict:
      return (out, has_nsfw)
    return PipelineOutput(images=out, nsfw_content_detected=has_nsfw)
