"""Build-owned deterministic synthetic weights and inputs.

No checkpoint of the reference is available offline (pretrained_ckpt.pth.tar, OPT, SD-1.5 are all
absent), so parity and benchmarks run on random weights of the exact shapes / state-dict names:
  * OPTForCausalLM   (transformers)           -> opt_state_dict
  * TextFcLayer 'gill_mapper' (gill/layers.py:17-24) -> mapper_state_dict
  * UNet2DConditionModel (diffusers, SD-1.5 config)  -> unet_state_dict
Values come from a counter-based generator (numpy Philox keyed by crc32(tensor name) and a seed), so
the same tensors are produced on any machine, independent of torch's RNG.  Matrices are fan-in scaled
(std = gain / sqrt(fan_in)) so activations stay O(1) through the whole path and the parity tests are
not trivially dominated by the inputs; norm scales/shifts are non-trivial on purpose.
"""
from __future__ import annotations

import zlib
from dataclasses import dataclass, field
from typing import Optional, Dict, List, Tuple

import numpy as np
import torch


def _rng(name: str, seed: int) -> np.random.Generator:
  return np.random.Generator(np.random.Philox(key=[zlib.crc32(name.encode()) & 0xFFFFFFFF, seed & 0xFFFFFFFF]))


def normal(name: str, shape, seed: int, std: float = 1.0, mean: float = 0.0) -> torch.Tensor:
  a = _rng(name, seed).standard_normal(tuple(shape), dtype=np.float32)
  if std != 1.0:
    a *= np.float32(std)
  if mean != 0.0:
    a += np.float32(mean)
  return torch.from_numpy(a)


def _matrix(name, shape, seed, gain=1.0):
  fan_in = int(np.prod(shape[1:]))
  return normal(name, shape, seed, gain / np.sqrt(fan_in))


def _bias(name, n, seed):
  return normal(name, (n,), seed, 0.02)


def _norm(sd, prefix, n, seed):
  sd[prefix + ".weight"] = normal(prefix + ".weight", (n,), seed, 0.1, 1.0)
  sd[prefix + ".bias"] = normal(prefix + ".bias", (n,), seed, 0.1)


def _linear(sd, prefix, out, inp, seed, bias=True, gain=1.0):
  sd[prefix + ".weight"] = _matrix(prefix + ".weight", (out, inp), seed, gain)
  if bias:
    sd[prefix + ".bias"] = _bias(prefix + ".bias", out, seed)


# ---------------------------------------------------------------------------------------------- OPT
@dataclass
class OptConfig:
  vocab_size: int = 50274      # 50265 + <|image|> + 8 [IMG] tokens (gill/models.py:845-862)
  hidden_size: int = 4096
  num_layers: int = 32
  num_heads: int = 32
  ffn_dim: int = 16384
  max_positions: int = 2048

  @staticmethod
  def opt_6_7b():
    return OptConfig()

  @staticmethod
  def opt_125m():
    return OptConfig(hidden_size=768, num_layers=12, num_heads=12, ffn_dim=3072)

  @staticmethod
  def tiny(vocab=512):
    return OptConfig(vocab_size=vocab, hidden_size=128, num_layers=2, num_heads=2, ffn_dim=256, max_positions=128)


def opt_state_dict(cfg: OptConfig, seed: int = 0) -> Dict[str, torch.Tensor]:
  sd: Dict[str, torch.Tensor] = {}
  D, F = cfg.hidden_size, cfg.ffn_dim
  sd["model.decoder.embed_tokens.weight"] = normal("model.decoder.embed_tokens.weight", (cfg.vocab_size, D), seed, 0.5)
  sd["model.decoder.embed_positions.weight"] = normal("model.decoder.embed_positions.weight", (cfg.max_positions + 2, D),
                                                      seed, 0.25)
  _norm(sd, "model.decoder.final_layer_norm", D, seed)
  for i in range(cfg.num_layers):
    p = f"model.decoder.layers.{i}"
    for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
      _linear(sd, f"{p}.self_attn.{n}", D, D, seed)
    _norm(sd, f"{p}.self_attn_layer_norm", D, seed)
    _linear(sd, f"{p}.fc1", F, D, seed)
    _linear(sd, f"{p}.fc2", D, F, seed)
    _norm(sd, f"{p}.final_layer_norm", D, seed)
  return sd


# ---------------------------------------------------------------------------------------------- CLIP vision tower
@dataclass
class ClipConfig:
  image_size: int = 224
  patch_size: int = 14
  hidden_size: int = 1024
  num_layers: int = 24
  num_heads: int = 16
  intermediate_size: int = 4096

  @staticmethod
  def vit_l14():
    return ClipConfig()

  @staticmethod
  def tiny():          # head dim 64 (a supported attention width), 5 tokens
    return ClipConfig(image_size=32, patch_size=16, hidden_size=128, num_layers=2, num_heads=2, intermediate_size=256)


def clip_state_dict(cfg: ClipConfig, seed: int = 0) -> Dict[str, torch.Tensor]:
  """transformers.CLIPVisionModel parameter names (the vision tower GILLModel holds as `visual_model`)."""
  sd: Dict[str, torch.Tensor] = {}
  D, F, P = cfg.hidden_size, cfg.intermediate_size, cfg.patch_size
  ntok = (cfg.image_size // P) ** 2 + 1
  vm = "vision_model"
  sd[f"{vm}.embeddings.class_embedding"] = normal(f"{vm}.embeddings.class_embedding", (D,), seed, 0.5)
  sd[f"{vm}.embeddings.patch_embedding.weight"] = _matrix(f"{vm}.embeddings.patch_embedding.weight", (D, 3, P, P), seed)
  sd[f"{vm}.embeddings.position_embedding.weight"] = normal(f"{vm}.embeddings.position_embedding.weight", (ntok, D), seed, 0.25)
  _norm(sd, f"{vm}.pre_layrnorm", D, seed)       # (sic: the Hugging Face name)
  for i in range(cfg.num_layers):
    p = f"{vm}.encoder.layers.{i}"
    for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
      _linear(sd, f"{p}.self_attn.{n}", D, D, seed)
    _norm(sd, f"{p}.layer_norm1", D, seed)
    _linear(sd, f"{p}.mlp.fc1", F, D, seed)
    _linear(sd, f"{p}.mlp.fc2", D, F, seed)
    _norm(sd, f"{p}.layer_norm2", D, seed)
  _norm(sd, f"{vm}.post_layernorm", D, seed)
  return sd


# ---------------------------------------------------------------------------------------------- GILLMapper
@dataclass
class MapperConfig:
  in_dim: int = 4096
  out_dim: int = 768
  hidden_dim: int = 512
  num_heads: int = 4
  ffn_dim: int = 2048
  num_enc_layers: int = 4
  num_dec_layers: int = 4
  num_input_tokens: int = 8
  num_output_tokens: int = 77


def mapper_state_dict(cfg: MapperConfig, seed: int = 0) -> Dict[str, torch.Tensor]:
  sd: Dict[str, torch.Tensor] = {}
  Hd, F = cfg.hidden_dim, cfg.ffn_dim
  sd["query_embs"] = normal("query_embs", (1, cfg.num_output_tokens, Hd), seed)
  _linear(sd, "fc", Hd, cfg.in_dim, seed)

  def mha(p):
    sd[p + ".in_proj_weight"] = _matrix(p + ".in_proj_weight", (3 * Hd, Hd), seed)
    sd[p + ".in_proj_bias"] = _bias(p + ".in_proj_bias", 3 * Hd, seed)
    _linear(sd, p + ".out_proj", Hd, Hd, seed)

  for i in range(cfg.num_enc_layers):
    p = f"tfm.encoder.layers.{i}"
    mha(p + ".self_attn")
    _linear(sd, p + ".linear1", F, Hd, seed)
    _linear(sd, p + ".linear2", Hd, F, seed)
    _norm(sd, p + ".norm1", Hd, seed)
    _norm(sd, p + ".norm2", Hd, seed)
  _norm(sd, "tfm.encoder.norm", Hd, seed)
  for i in range(cfg.num_dec_layers):
    p = f"tfm.decoder.layers.{i}"
    mha(p + ".self_attn")
    mha(p + ".multihead_attn")
    _linear(sd, p + ".linear1", F, Hd, seed)
    _linear(sd, p + ".linear2", Hd, F, seed)
    for k in (1, 2, 3):
      _norm(sd, p + f".norm{k}", Hd, seed)
  _norm(sd, "tfm.decoder.norm", Hd, seed)
  _linear(sd, "model", cfg.out_dim, Hd, seed)
  return sd


# ---------------------------------------------------------------------------------------------- SD UNet
@dataclass
class UNetConfig:
  in_channels: int = 4
  out_channels: int = 4
  block_out_channels: Tuple[int, int, int, int] = (320, 640, 1280, 1280)
  layers_per_block: int = 2
  cross_attention_dim: int = 768
  num_heads: int = 8           # SD-1.5's config calls this "attention_head_dim": it is the head COUNT
  norm_num_groups: int = 32
  sample_size: int = 64
  ctx_len: int = 77
  heads_per_level: Optional[Tuple[int, int, int, int]] = None   # SD-2.x: fixed head dim 64 -> (5, 10, 20, 20) heads
  prediction_type: str = "epsilon"                              # "v_prediction" for SD-2.1-768
  fp8_convs: bool = False      # BASELINE.json configs[4]: resnet convolutions in fp8 e4m3 (csrc/conv_fp8.hip); not a parity mode

  def heads(self, level: int) -> int:
    return self.heads_per_level[level] if self.heads_per_level else self.num_heads

  @staticmethod
  def sd15():
    return UNetConfig()

  @staticmethod
  def sd21_768():
    """BASELINE.json configs[3]: the SD-2.1 768x768 UNet (same topology; head dim 64 at every level, OpenCLIP 1024-d
    context, 96x96 latents, v-prediction; its Linear proj_in/proj_out are the same maps as SD-1.x's 1x1 convs)."""
    return UNetConfig(cross_attention_dim=1024, sample_size=96, heads_per_level=(5, 10, 20, 20), prediction_type="v_prediction")

  @staticmethod
  def tiny_sd2(sample_size=16):
    return UNetConfig(block_out_channels=(64, 128, 256, 256), cross_attention_dim=128, sample_size=sample_size,
                      heads_per_level=(1, 2, 4, 4), prediction_type="v_prediction")

  @staticmethod
  def tiny(sample_size=16):
    # same topology, 1/5 width, 4 heads: head dims 16/32/64/64 -> padded 48/48/64/64 (heads*dp % 64 == 0)
    return UNetConfig(block_out_channels=(64, 128, 256, 256), num_heads=4, cross_attention_dim=128,
                      sample_size=sample_size)


def unet_state_dict(cfg: UNetConfig, seed: int = 0) -> Dict[str, torch.Tensor]:
  sd: Dict[str, torch.Tensor] = {}
  ch = cfg.block_out_channels
  temb = ch[0] * 4

  def conv(p, cout, cin, k=3):
    sd[p + ".weight"] = _matrix(p + ".weight", (cout, cin, k, k), seed)
    sd[p + ".bias"] = _bias(p + ".bias", cout, seed)

  def resnet(p, cin, cout):
    _norm(sd, p + ".norm1", cin, seed)
    conv(p + ".conv1", cout, cin)
    _linear(sd, p + ".time_emb_proj", cout, temb, seed)
    _norm(sd, p + ".norm2", cout, seed)
    conv(p + ".conv2", cout, cout)
    if cin != cout:
      conv(p + ".conv_shortcut", cout, cin, 1)

  def xf(p, C):
    _norm(sd, p + ".norm", C, seed)
    conv(p + ".proj_in", C, C, 1)
    b = p + ".transformer_blocks.0"
    for k in (1, 2, 3):
      _norm(sd, b + f".norm{k}", C, seed)
    for a, kvd in (("attn1", C), ("attn2", cfg.cross_attention_dim)):
      _linear(sd, f"{b}.{a}.to_q", C, C, seed, bias=False)
      _linear(sd, f"{b}.{a}.to_k", C, kvd, seed, bias=False)
      _linear(sd, f"{b}.{a}.to_v", C, kvd, seed, bias=False)
      _linear(sd, f"{b}.{a}.to_out.0", C, C, seed)
    _linear(sd, b + ".ff.net.0.proj", 8 * C, C, seed)
    _linear(sd, b + ".ff.net.2", C, 4 * C, seed)
    conv(p + ".proj_out", C, C, 1)

  conv("conv_in", ch[0], cfg.in_channels)
  _linear(sd, "time_embedding.linear_1", temb, ch[0], seed)
  _linear(sd, "time_embedding.linear_2", temb, temb, seed)
  for i in range(4):
    cin = ch[0] if i == 0 else ch[i - 1]
    for j in range(2):
      resnet(f"down_blocks.{i}.resnets.{j}", cin if j == 0 else ch[i], ch[i])
      if i < 3:
        xf(f"down_blocks.{i}.attentions.{j}", ch[i])
    if i < 3:
      conv(f"down_blocks.{i}.downsamplers.0.conv", ch[i], ch[i])
  resnet("mid_block.resnets.0", ch[3], ch[3])
  xf("mid_block.attentions.0", ch[3])
  resnet("mid_block.resnets.1", ch[3], ch[3])
  rev = (ch[3], ch[2], ch[1], ch[0])
  for i in range(4):
    outc = rev[i]
    prev = rev[0] if i == 0 else rev[i - 1]
    inc = rev[min(i + 1, 3)]
    for j in range(3):
      skip = inc if j == 2 else outc
      rin = prev if j == 0 else outc
      resnet(f"up_blocks.{i}.resnets.{j}", rin + skip, outc)
      if i > 0:
        xf(f"up_blocks.{i}.attentions.{j}", outc)
    if i < 3:
      conv(f"up_blocks.{i}.upsamplers.0.conv", outc, outc)
  _norm(sd, "conv_norm_out", ch[0], seed)
  conv("conv_out", cfg.out_channels, ch[0])
  return sd


# ---------------------------------------------------------------------------------------------- SD VAE (decoder half)
@dataclass
class VAEConfig:
  latent_channels: int = 4
  out_channels: int = 3
  block_out_channels: Tuple[int, int, int, int] = (128, 256, 512, 512)
  layers_per_block: int = 2
  norm_num_groups: int = 32
  latent_size: int = 64
  scaling_factor: float = 0.18215

  @staticmethod
  def sd15():
    return VAEConfig()

  @staticmethod
  def tiny(latent_size=16):
    return VAEConfig(block_out_channels=(64, 64, 128, 128), latent_size=latent_size)


def vae_decoder_state_dict(cfg: VAEConfig, seed: int = 0) -> Dict[str, torch.Tensor]:
  """diffusers AutoencoderKL keys needed by decode(): post_quant_conv.* and decoder.*"""
  sd: Dict[str, torch.Tensor] = {}
  ch = cfg.block_out_channels

  def conv(p, cout, cin, k=3):
    sd[p + ".weight"] = _matrix(p + ".weight", (cout, cin, k, k), seed)
    sd[p + ".bias"] = _bias(p + ".bias", cout, seed)

  def resnet(p, cin, cout):
    _norm(sd, p + ".norm1", cin, seed)
    conv(p + ".conv1", cout, cin)
    _norm(sd, p + ".norm2", cout, seed)
    conv(p + ".conv2", cout, cout)
    if cin != cout:
      conv(p + ".conv_shortcut", cout, cin, 1)

  conv("post_quant_conv", cfg.latent_channels, cfg.latent_channels, 1)
  top = ch[3]
  conv("decoder.conv_in", top, cfg.latent_channels)
  resnet("decoder.mid_block.resnets.0", top, top)
  a = "decoder.mid_block.attentions.0"
  _norm(sd, a + ".group_norm", top, seed)
  for n in ("to_q", "to_k", "to_v", "to_out.0"):
    _linear(sd, f"{a}.{n}", top, top, seed)
  resnet("decoder.mid_block.resnets.1", top, top)
  prev = top
  for i in range(4):
    outc = ch[3 - i]
    for j in range(3):
      resnet(f"decoder.up_blocks.{i}.resnets.{j}", prev if j == 0 else outc, outc)
    if i < 3:
      conv(f"decoder.up_blocks.{i}.upsamplers.0.conv", outc, outc)
    prev = outc
  _norm(sd, "decoder.conv_norm_out", ch[0], seed)
  conv("decoder.conv_out", cfg.out_channels, ch[0])
  return sd


# ---------------------------------------------------------------------------------------------- inputs
IMG_TOKEN_IDS = list(range(50266, 50274))  # checkpoints/gill_opt/model_args.json:19-36


def synthetic_prompt_ids(batch: int, prompt_len: int = 24, seed: int = 0, vocab_lo: int = 3, vocab_hi: int = 50264,
                         img_ids: List[int] = IMG_TOKEN_IDS, bos: int = 2) -> torch.Tensor:
  """(B, prompt_len + 8) int64: BOS, uniform-random word ids, then [IMG0..7]."""
  g = _rng("prompt_ids", seed)
  ids = g.integers(vocab_lo, vocab_hi + 1, size=(batch, prompt_len), dtype=np.int64)
  ids[:, 0] = bos
  img = np.tile(np.asarray(img_ids, dtype=np.int64)[None, :], (batch, 1))
  return torch.from_numpy(np.concatenate([ids, img], axis=1))


def initial_latents(batch: int, channels: int = 4, size: int = 64, seed: int = 1337) -> torch.Tensor:
  """fp32 (B,4,L,L) from torch.Generator('cpu').manual_seed(seed + i) per prompt (not the CUDA Philox stream)."""
  outs = []
  for i in range(batch):
    g = torch.Generator(device="cpu").manual_seed(seed + i)
    outs.append(torch.randn((channels, size, size), generator=g, dtype=torch.float32))
  return torch.stack(outs, 0)


def uncond_context(ctx_len: int = 77, dim: int = 768, seed: int = 0) -> torch.Tensor:
  """Stand-in for CLIP-text('') (unobtainable offline): fixed random (1,77,768)."""
  return normal("uncond_context", (1, ctx_len, dim), seed)


def host_cores() -> int:
  """CPU cores this process may actually use: min(affinity mask, cgroup cpu.max quota).  (The MI355X boxes expose 256
  logical CPUs but run under a 16-CPU quota; oversubscribing torch's CPU thread pool there is ~20x slower.)"""
  import os
  n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
  try:
    quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
    if quota != "max":
      n = min(n, max(1, int(int(quota) / int(period))))
  except Exception:
    pass
  return max(1, n)


# ---------------------------------------------------------------------------------------------- tokenizer stand-in
class HashTokenizer:
  """Offline stand-in for the OPT GPT2 tokenizer (no vocab files exist in this environment).  Exposes the
  attributes/methods the reference touches (gill/models.py:44, :221, :615, :636, :657, :708; load_gill :845-862):
  whitespace words hash to ids in [3, 50264]; '[IMGk]' -> 50266+k; '<|image|>' -> 50265; '\\n' -> 50118."""
  bos_token_id = 2
  eos_token_id = 2
  pad_token_id = 1
  cls_token_id = 50265
  newline_id = 50118

  def __init__(self, num_img_tokens: int = 8):
    self.num_img_tokens = num_img_tokens

  def __len__(self):
    return 50266 + self.num_img_tokens

  def _encode(self, text: str):
    import re
    ids = []
    for tok in re.findall(r"\[IMG\d+\]|<\|image\|>|\n|[^\s\[\n]+|\[", text):
      m = re.fullmatch(r"\[IMG(\d+)\]", tok)
      if m:
        ids.append(50266 + int(m.group(1)))
      elif tok == "<|image|>":
        ids.append(self.cls_token_id)
      elif tok == "\n":
        ids.append(self.newline_id)
      else:
        ids.append(3 + (zlib.crc32(tok.encode()) % 50000))
    return ids

  def __call__(self, text, add_special_tokens: bool = True, return_tensors=None, **_):
    from types import SimpleNamespace
    ids = self._encode(text)
    if add_special_tokens:
      ids = [self.bos_token_id] + ids
    if return_tensors == "pt":
      return SimpleNamespace(input_ids=torch.tensor([ids], dtype=torch.int64))
    return SimpleNamespace(input_ids=ids)

  def batch_decode(self, ids, skip_special_tokens: bool = True):
    out = []
    for row in ids:
      words = []
      for t in (row.tolist() if hasattr(row, "tolist") else row):
        if skip_special_tokens and (t in (self.bos_token_id, self.pad_token_id) or t >= 50265):
          continue
        words.append("\n" if t == self.newline_id else f"w{t}")
      out.append(" ".join(words))
    return out
