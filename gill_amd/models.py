"""Mirror of the reference's public surface gill/models.py — GILLArgs, GILLModel, GILL, load_gill — for the
image-generation hot path, with every FLOP in libgill_amd (hand-written HIP for MI355X).

Reference map (file:line in /root/reference/gill/models.py):
  GILLArgs                                   :21-36
  GILLModel.__init__ / get_visual_embs       :40-126 / :129-152
  GILLModel.forward(mode='generation')       :164-441  (relevant: :173-193, :276-299, :357-365, :374-387, :416-419)
  GILLModel.generate                         :443-532
  GILL.__init__ / __call__                   :536-561 / :563-580
  GILL.generate_for_images_and_texts         :582-762  ('gen' branch: :600-662, :706-731, :754-762)
  load_gill                                  :810-902
New here (NOT in the reference): GILL.generate_images — the batched entry whose
per-prompt semantics are the 'gen' branch of generate_for_images_and_texts([p], num_words=2, gen_scale_factor=1e5).

Also built: image prompts through the native CLIP vision tower (get_visual_embs), KV-cached decoding in
generate(), the retrieval / decision / CLIP-rerank branches of generate_for_images_and_texts, load_gill with the cc3m*.npy
retrieval embeddings.  Out of scope (raise NotImplementedError): the captioning / retrieval TRAINING modes of
GILLModel.forward.
"""
from __future__ import annotations

import ctypes as C
import glob
import json
import os
from collections import namedtuple
from types import SimpleNamespace
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch import Tensor

from . import _native as N
from . import layers, utils
from .synth import ClipConfig, OptConfig


class GILLArgs:
  freeze_lm: bool = True
  freeze_vm: bool = True
  opt_version: str = 'facebook/opt-6.7b'
  visual_encoder: str = 'openai/clip-vit-large-patch14'
  n_visual_tokens: int = 1
  task: str = 'captioning'
  ret_emb_dim: Optional[int] = 256
  gen_emb_dim: Optional[int] = 256
  text_emb_layers: List[int] = [-1]
  gen_token_idx: List[int] = [0]
  retrieval_token_idx: List[int] = [0]
  text_fc_mode: str = 'gill_mapper'
  ret_text_fc_mode: str = 'linear'
  num_tokens: int = 8
  num_clip_tokens: int = 77


_OPT_SHAPES = {  # hidden, layers, heads, ffn  (public OPT configs; 350m is post-LN + project_in/out: unsupported)
  'opt-125m': (768, 12, 12, 3072), 'opt-1.3b': (2048, 24, 32, 8192), 'opt-2.7b': (2560, 32, 32, 10240),
  'opt-6.7b': (4096, 32, 32, 16384), 'opt-13b': (5120, 40, 40, 20480),
}
_CLIP_HIDDEN = {'clip-vit-large-patch14': 1024, 'clip-vit-base-patch16': 768, 'clip-vit-base-patch32': 768}


class _ParamTree(nn.Module):
  """Parameter container whose state_dict() keys reproduce a given set of dotted names (so a module tree of
  transformers' OPTForCausalLM can be held — and checkpoint-loaded — without running any of its code)."""

  def __init__(self, state: Dict[str, Tensor]):
    super().__init__()
    for name, t in state.items():
      mod = self
      parts = name.split('.')
      for p in parts[:-1]:
        if p not in mod._modules:
          mod.add_module(p, _ParamTree({}))
        mod = mod._modules[p]
      mod.register_parameter(parts[-1], nn.Parameter(t, requires_grad=False))


class _NativeEmbedding(nn.Module):
  """`input_embeddings`: shares the OPT token-embedding parameter; lookup runs in libgill_amd (gill_opt_embed)."""

  def __init__(self, owner: "GILLModel", weight: nn.Parameter):
    super().__init__()
    self.weight = weight
    self._owner = [owner]   # list: keep the back-reference out of nn.Module's child registry
    self.embedding_dim = weight.shape[1]
    self.num_embeddings = weight.shape[0]

  def forward(self, ids: Tensor) -> Tensor:
    owner = self._owner[0]
    h = owner._opt_native(1, 1)
    ids = ids.to(self.weight.device, torch.int64).contiguous()
    out = torch.empty(tuple(ids.shape) + (self.embedding_dim,), device=self.weight.device, dtype=torch.bfloat16)
    with torch.cuda.device(self.weight.device):
      N.check(N.lib().gill_opt_embed(h, N.ptr(ids), ids.numel(), N.ptr(out), N.current_stream()))
    return out.to(self.weight.dtype)


class GILLModel(nn.Module):
  def __init__(self, tokenizer, args: GILLArgs = GILLArgs()):
    super().__init__()
    self.tokenizer = tokenizer
    self.feature_extractor = utils.get_feature_extractor_for_model(args.visual_encoder, train=False)   # models.py:48
    self.image_token = self.tokenizer.cls_token_id
    assert args.text_emb_layers != set(args.text_emb_layers), 'text_emb_layers not unique'
    self.args = args
    self.num_tokens = args.num_tokens
    self.num_clip_tokens = args.num_clip_tokens

    opt_version = args.opt_version
    visual_encoder = args.visual_encoder
    n_visual_tokens = args.n_visual_tokens
    print(f"Using {opt_version} for the language model.")
    print(f"Using {visual_encoder} for the visual model with {n_visual_tokens} visual tokens.")

    if 'facebook/opt' not in opt_version:
      raise NotImplementedError
    self.opt_version = opt_version
    self.opt_cfg, lm_state = self._load_opt(opt_version, len(tokenizer), getattr(args, 'opt_state_dict', None))
    self.lm = _ParamTree(lm_state)
    self.lm.config = SimpleNamespace(word_embed_proj_dim=self.opt_cfg.hidden_size, hidden_size=self.opt_cfg.hidden_size,
                                     num_hidden_layers=self.opt_cfg.num_layers)
    print("Freezing the LM.")   # the LM is always frozen here: this package is inference-only

    self.retrieval_token_idx = args.retrieval_token_idx
    self.gen_token_idx = args.gen_token_idx
    self.input_embeddings = _NativeEmbedding(self, self.lm.model.decoder.embed_tokens.weight)

    self.visual_model_name = visual_encoder
    # frozen CLIP vision tower (models.py:78-96): a parameter container; its forward runs in libgill_amd (gill_clip_*).
    # Weights: args.clip_state_dict (synthetic / pre-loaded), a local Hugging Face directory, or absent — then text prompts
    # work as always and image prompts raise.
    self.clip_cfg, clip_state = self._load_clip(visual_encoder, getattr(args, 'clip_state_dict', None),
                                                getattr(args, 'clip_config', None))
    self.visual_model = _ParamTree(clip_state) if clip_state is not None else None
    hidden_size = self.clip_cfg.hidden_size if self.clip_cfg is not None else self._clip_hidden(visual_encoder)
    if self.clip_cfg is not None and self.clip_cfg.image_size != self.feature_extractor.size:
      self.feature_extractor = utils.ClipImageProcessor(self.clip_cfg.image_size, self.clip_cfg.image_size)
    self._clip_handle = None

    embedding_dim = self.input_embeddings.embedding_dim * self.args.n_visual_tokens
    self.ret_text_hidden_fcs = nn.ModuleList([])
    self.gen_text_hidden_fcs = nn.ModuleList([])
    for layer_idx in self.args.text_emb_layers:
      if (layer_idx == -1 or layer_idx == self.lm.config.num_hidden_layers) and ('bert' not in opt_version):
        in_dim = self.lm.config.word_embed_proj_dim
      elif layer_idx < self.lm.config.num_hidden_layers:
        raise NotImplementedError('only text_emb_layers=[-1] (the shipped configuration) is supported')
      else:
        raise ValueError(f'Embedding of layer {layer_idx} was requested but model only has {self.lm.config.num_hidden_layers} layers.')
      self.ret_text_hidden_fcs.append(
        layers.TextFcLayer(in_dim, self.args.ret_emb_dim, num_input_tokens=self.args.num_tokens,
                           num_output_tokens=1, mode=self.args.ret_text_fc_mode))
      self.gen_text_hidden_fcs.append(
        layers.TextFcLayer(in_dim, self.args.gen_emb_dim, num_input_tokens=self.args.num_tokens,
                           num_output_tokens=self.args.num_clip_tokens, mode=self.args.text_fc_mode))

    # parameters of the out-of-scope branches are kept so reference checkpoints load with matching keys
    self.visual_embeddings = nn.Linear(hidden_size, embedding_dim)
    self.visual_fc = nn.Linear(hidden_size, self.args.ret_emb_dim)
    self.logit_scale = nn.Parameter(torch.ones([]) * np.log(1 / 0.07))
    self._opt_handle = None
    self._opt_cap = (0, 0)
    # native handles snapshot the weights: a state-dict load into this module (or, recursively, into a parent) drops them
    self.register_load_state_dict_post_hook(lambda module, incompatible_keys: module.refresh_native())

  # ---- weights -------------------------------------------------------------------------------
  @staticmethod
  def _clip_hidden(name: str) -> int:
    for k, v in _CLIP_HIDDEN.items():
      if k in name:
        return v
    raise NotImplementedError(f'unknown visual encoder {name}')

  @staticmethod
  def _load_opt(opt_version: str, vocab: int, state: Optional[Dict[str, Tensor]]):
    """Returns (OptConfig, OPTForCausalLM-named state dict with embeddings resized to `vocab` (models.py:73))."""
    if state is not None:   # caller-supplied weights (synthetic weights for tests / benchmarks)
      D = state['model.decoder.embed_tokens.weight'].shape[1]
      n_layers = 1 + max(int(k.split('.')[3]) for k in state if k.startswith('model.decoder.layers.'))
      F = state['model.decoder.layers.0.fc1.weight'].shape[0]
      shape = [v for k, v in _OPT_SHAPES.items() if k in opt_version]
      heads = shape[0][2] if shape and shape[0][0] == D else max(1, D // 64)
      heads = getattr(state, 'num_heads', heads)
      cfg = OptConfig(vocab_size=vocab, hidden_size=D, num_layers=n_layers, num_heads=heads, ffn_dim=F,
                      max_positions=state['model.decoder.embed_positions.weight'].shape[0] - 2)
      sd = dict(state)
    else:                   # the reference's way: transformers is the weight loader (never the executor)
      from transformers import OPTForCausalLM
      hf = OPTForCausalLM.from_pretrained(opt_version)
      c = hf.config
      if not c.do_layer_norm_before or c.word_embed_proj_dim != c.hidden_size:
        raise NotImplementedError('post-LN / projected OPT variants (opt-350m) are not supported')
      cfg = OptConfig(vocab_size=vocab, hidden_size=c.hidden_size, num_layers=c.num_hidden_layers,
                      num_heads=c.num_attention_heads, ffn_dim=c.ffn_dim, max_positions=c.max_position_embeddings)
      sd = {k: v for k, v in hf.state_dict().items() if k != 'lm_head.weight'}   # tied to embed_tokens
    emb = sd['model.decoder.embed_tokens.weight']
    if emb.shape[0] != vocab:   # resize_token_embeddings(len(tokenizer)): new rows ~ N(0, 0.02) like HF's _init_weights
      new = torch.empty((vocab, emb.shape[1]), dtype=emb.dtype).normal_(0.0, 0.02)
      n = min(vocab, emb.shape[0])
      new[:n] = emb[:n]
      sd['model.decoder.embed_tokens.weight'] = new
    return cfg, sd

  @staticmethod
  def _load_clip(name: str, state: Optional[Dict[str, Tensor]], cfg: Optional[ClipConfig]):
    """(ClipConfig, CLIPVisionModel-named state dict) or (None, None) when no vision weights are available offline."""
    if state is not None:
      if cfg is None:
        D = state['vision_model.embeddings.class_embedding'].shape[0]
        P = state['vision_model.embeddings.patch_embedding.weight'].shape[-1]
        ntok = state['vision_model.embeddings.position_embedding.weight'].shape[0]
        n_layers = 1 + max(int(k.split('.')[3]) for k in state if k.startswith('vision_model.encoder.layers.'))
        F = state['vision_model.encoder.layers.0.mlp.fc1.weight'].shape[0]
        cfg = ClipConfig(image_size=int(round((ntok - 1) ** 0.5)) * P, patch_size=P, hidden_size=D, num_layers=n_layers,
                         num_heads=max(1, D // 64), intermediate_size=F)
      return cfg, dict(state)
    if os.path.isdir(name) and os.path.exists(os.path.join(name, 'config.json')):   # local HF directory
      from transformers import CLIPVisionModel
      hf = CLIPVisionModel.from_pretrained(name)
      c = hf.config
      sd = {(k if k.startswith('vision_model.') else 'vision_model.' + k): v for k, v in hf.state_dict().items()}
      return ClipConfig(image_size=c.image_size, patch_size=c.patch_size, hidden_size=c.hidden_size,
                        num_layers=c.num_hidden_layers, num_heads=c.num_attention_heads,
                        intermediate_size=c.intermediate_size), sd
    return None, None

  def _clip_native(self, B: int):
    dev = self.logit_scale.device
    if dev.type != 'cuda':
      raise N.GillNativeError('GILLModel runs only on an MI355X through libgill_amd; call .cuda() first.')
    if self.visual_model is None:
      raise NotImplementedError('image prompts need the CLIP vision weights (args.clip_state_dict or a local '
                                'openai/clip-vit-* directory): none were available when this model was built')
    if self._clip_handle is not None and B <= self._clip_cap:
      return self._clip_handle
    if self._clip_handle is not None:
      N.lib().gill_clip_destroy(self._clip_handle)
      self._clip_handle = None
    c = self.clip_cfg
    cap = max(B, 4)
    cfg = N.gill_clip_config(image_size=c.image_size, patch_size=c.patch_size, hidden_size=c.hidden_size,
                             num_layers=c.num_layers, num_heads=c.num_heads, intermediate_size=c.intermediate_size,
                             max_batch=cap)
    arr, keep = N.make_tensor_table(self.visual_model.state_dict(), dev)
    h = C.c_void_p()
    with torch.cuda.device(dev):
      N.check(N.lib().gill_clip_create(C.byref(h), C.byref(cfg), arr, len(keep)))
    del keep
    self._clip_handle, self._clip_cap = h, cap
    return h

  def _apply(self, fn, *a, **k):
    self.release_native()
    return super()._apply(fn, *a, **k)

  def release_native(self):
    if getattr(self, '_opt_handle', None):
      N.lib().gill_opt_destroy(self._opt_handle)
    self._opt_handle = None
    self._opt_cap = (0, 0)
    if getattr(self, '_clip_handle', None):
      N.lib().gill_clip_destroy(self._clip_handle)
    self._clip_handle = None

  def refresh_native(self):
    """Call after mutating weights in place once a forward has already run (handles snapshot the weights)."""
    self.release_native()
    for fcs in (self.gen_text_hidden_fcs, getattr(self, 'ret_text_hidden_fcs', [])):
      for fc in fcs:
        if hasattr(fc, 'release_native'):
          fc.release_native()

  def __del__(self):
    try:
      self.release_native()
    except Exception:
      pass

  def _opt_native(self, B: int, T: int):
    dev = self.logit_scale.device
    if dev.type != 'cuda':
      raise N.GillNativeError('GILLModel runs only on an MI355X through libgill_amd; call .cuda() first '
                              '(there is no CPU implementation in this package).')
    cb, ct = self._opt_cap
    if self._opt_handle is not None and B <= cb and T <= ct:
      return self._opt_handle
    self.release_native()
    cb, ct = max(cb, B, 8), max(ct, T, 64)
    c = self.opt_cfg
    cfg = N.gill_opt_config(vocab_size=c.vocab_size, hidden_size=c.hidden_size, num_layers=c.num_layers,
                            num_heads=c.num_heads, ffn_dim=c.ffn_dim, max_positions=c.max_positions, max_batch=cb, max_seq=ct)
    arr, keep = N.make_tensor_table(self.lm.state_dict(), dev)
    h = C.c_void_p()
    with torch.cuda.device(dev):
      N.check(N.lib().gill_opt_create(C.byref(h), C.byref(cfg), arr, len(keep)))
    del keep
    self._opt_handle, self._opt_cap = h, (cb, ct)
    return h

  # ---- reference API -------------------------------------------------------------------------
  def get_visual_embs(self, pixel_values: torch.FloatTensor, mode: str = 'captioning'):
    if mode not in ['captioning', 'retrieval', 'generation']:
      raise ValueError(f"mode should be one of ['captioning', 'retrieval', 'generation'], got {mode} instead.")
    if mode == 'generation':   # models.py:147-148: the image is ignored
      return torch.zeros((pixel_values.shape[0], 1, 768), device=pixel_values.device)
    # models.py:137-146: pooler_output of the frozen tower -> visual_embeddings / visual_fc -> (B, n_visual_tokens | 1, D)
    from . import ops
    px = pixel_values.to(self.logit_scale.device, torch.float32).contiguous()
    B = px.shape[0]
    h = self._clip_native(B)
    pooled = torch.empty((B, self.clip_cfg.hidden_size), device=px.device, dtype=torch.float32)
    with torch.cuda.device(px.device):
      N.check(N.lib().gill_clip_forward(h, N.ptr(px), B, N.ptr(pooled), N.current_stream()))
    fc = self.visual_embeddings if mode == 'captioning' else self.visual_fc
    out = ops.gemm(pooled.to(torch.bfloat16), fc.weight, bias=fc.bias, out_f32=True)
    n_tok = self.args.n_visual_tokens if mode == 'captioning' else 1
    return out.reshape(B, n_tok, -1).to(self.logit_scale.dtype)

  def train(self, mode=True):
    super(GILLModel, self).train(mode=mode)   # reference quirk kept: returns None (models.py:155-161)

  def _lm_forward_hidden(self, inputs_embeds: Tensor) -> Tensor:
    """self.lm(inputs_embeds=..., output_hidden_states=True).hidden_states[-1]  (models.py:363-365 / :465)."""
    B, T, D = inputs_embeds.shape
    h = self._opt_native(B, T)
    x = inputs_embeds.to(torch.bfloat16).contiguous()
    out = torch.empty((B, T, D), device=x.device, dtype=torch.float32)
    with torch.cuda.device(x.device):
      N.check(N.lib().gill_opt_forward(h, N.ptr(x), B, T, N.ptr(out), N.current_stream()))
    return out

  def _lm_forward_hidden_cached(self, new_embeds: Tensor, past_len: int) -> Tensor:
    """hidden_states[-1] rows of the tokens past_len .. past_len+T_new-1, computed against the handle's KV cache
    (gill_opt_forward_cached).  past_len == 0 starts a new sequence."""
    B, Tn, D = new_embeds.shape
    cb, ct = self._opt_cap
    if past_len > 0 and (self._opt_handle is None or B > cb or past_len + Tn > ct):
      # growing would rebuild the handle and lose the keys / values cached so far
      raise N.GillNativeError(f'KV cache of the OPT handle holds {ct} tokens x {cb} sequences; step needs {past_len + Tn} x {B}. '
                              f'generate() sizes it up front (max_positions = {self.opt_cfg.max_positions}).')
    h = self._opt_native(B, past_len + Tn)
    x = new_embeds.to(torch.bfloat16).contiguous()
    out = torch.empty((B, Tn, D), device=x.device, dtype=torch.float32)
    with torch.cuda.device(x.device):
      N.check(N.lib().gill_opt_forward_cached(h, N.ptr(x), B, Tn, past_len, N.ptr(out), N.current_stream()))
    return out

  def img_hidden_states(self, labels: Tensor, last_embedding_idx: Tensor):
    """Fast path of forward(mode='generation'): (B,T) ids, (B,) idx -> (raw (B,8,D), embs (B,8,D)) bf16."""
    B, T = labels.shape
    dev = self.logit_scale.device
    h = self._opt_native(B, T)
    ids = labels.to(dev, torch.int64).contiguous()
    li = [int(v) for v in last_embedding_idx.reshape(-1).tolist()]
    arr = (C.c_int32 * B)(*li)
    D = self.opt_cfg.hidden_size
    raw = torch.empty((B, self.num_tokens, D), device=dev, dtype=torch.bfloat16)
    emb = torch.empty((B, self.num_tokens, D), device=dev, dtype=torch.bfloat16)
    with torch.cuda.device(dev):
      N.check(N.lib().gill_opt_img_hidden(h, N.ptr(ids), arr, B, T, self.num_tokens, N.ptr(raw), N.ptr(emb), N.current_stream()))
    return raw, emb

  def forward(self, pixel_values: torch.FloatTensor, labels: Optional[torch.LongTensor] = None,
              caption_len: Optional[torch.LongTensor] = None, mode: str = 'captioning', concat_captions: bool = False,
              input_prefix: Optional[str] = None):
    if mode != 'generation' or concat_captions or input_prefix is not None:
      raise NotImplementedError("only mode='generation' without concat_captions/input_prefix is on the inference hot path; "
                                "captioning / retrieval are training-time modes of the reference")
    visual_embs = self.get_visual_embs(pixel_values, mode)
    batch_size = visual_embs.shape[0]
    assert labels.shape[0] == batch_size, (visual_embs.shape, labels.shape)
    visual_embs_norm = ((visual_embs ** 2).sum(dim=-1) ** 0.5).mean()
    last_embedding_idx = caption_len - 1   # models.py:183

    # models.py:277-298, 358-362: label masking (loss bookkeeping only; vectorised, no effect on the embeddings)
    full_labels = torch.clone(labels)
    pad = self.tokenizer.pad_token_id
    stop = torch.zeros_like(full_labels, dtype=torch.bool)
    stop |= (full_labels == pad)
    for tok in (self.retrieval_token_idx[1:] + self.gen_token_idx[1:]):
      stop |= (full_labels == tok)
    full_labels = torch.where(stop.cumsum(dim=1) > 0, torch.full_like(full_labels, -100), full_labels)

    raw, emb = self.img_hidden_states(labels, last_embedding_idx)      # models.py:363-365, 384-385
    dt = self.logit_scale.dtype
    llm_hidden_states = [raw.to(dt)]
    hidden = [fc(raw.to(dt), emb.to(dt)) for fc in self.gen_text_hidden_fcs]   # models.py:387
    last_embedding = torch.stack(hidden, dim=-1).sum(dim=-1)                   # models.py:418
    input_embs_norm = None      # diagnostics of the training loop; not produced by the fast path
    output = SimpleNamespace(loss=None, logits=None, hidden_states=None)       # LM loss/logits are training-time outputs
    last_output_logit = None
    return output, full_labels, last_embedding, last_output_logit, visual_embs, visual_embs_norm, input_embs_norm, llm_hidden_states

  def generate(self, embeddings=torch.FloatTensor, max_len: int = 32, temperature: float = 0.0, top_p: float = 1.0,
               min_word_tokens: int = 0, ret_scale_factor: float = 1.0, gen_scale_factor: float = 1.0,
               filter_value: float = -float('Inf'), use_kv_cache: bool = True):
    """Greedy / top-p decoding, same loop and outputs as the reference (models.py:443-532).  The reference re-runs the
    whole sequence through the LM at every step; with use_kv_cache (default) only the tokens appended since the last
    step go through the layers, against keys/values cached in the native handle — causal attention makes the hidden
    states of earlier positions independent of later tokens, so the outputs are the same up to rounding.
    Outputs: out (N,T) token ids, output_embeddings list of hidden_states[-1], output_logits list (N, vocab)."""
    with torch.no_grad():
      out = None
      output_embeddings = []
      output_logits = []
      dev = embeddings.device
      vocab = self.opt_cfg.vocab_size
      hidden = None
      if use_kv_cache:
        # the cache lives in the handle: size it for the longest sequence this call can produce — every one of the max_len
        # steps may emit [IMG0] and so append all len(retrieval_token_idx) forced tokens (:518-520) — capped at the LM's
        # position table.  The handle is never rebuilt mid-sequence (_lm_forward_hidden_cached raises instead).
        grow = max(1, len(self.retrieval_token_idx))
        self._opt_native(embeddings.shape[0], min(embeddings.shape[1] + max_len * grow, self.opt_cfg.max_positions))
      for i in range(max_len):
        if use_kv_cache:
          past = 0 if hidden is None else hidden.shape[1]
          new_hidden = self._lm_forward_hidden_cached(embeddings[:, past:], past)
          hidden = new_hidden if hidden is None else torch.cat([hidden, new_hidden], dim=1)
        else:
          hidden = self._lm_forward_hidden(embeddings)                         # :465
        for idx in self.args.text_emb_layers:
          output_embeddings.append(hidden.to(embeddings.dtype))                # :467-468
        B, T, D = hidden.shape
        if B > 8:
          raise NotImplementedError('generate(): batch <= 8')
        logits = torch.empty((B, vocab), device=dev, dtype=torch.float32)
        with torch.cuda.device(dev):
          N.check(N.lib().gill_opt_last_logits(self._opt_handle, N.ptr(hidden), B, T, N.ptr(logits), N.current_stream()))
        if top_p == 1.0:
          logits = logits.cpu()                                                # :471-472
        output_logits.append(logits)
        logits[:, self.retrieval_token_idx[1:]] = filter_value                 # :476-477
        logits[:, self.gen_token_idx[1:]] = filter_value
        if (self.retrieval_token_idx or self.gen_token_idx) and self.retrieval_token_idx[0] != -1 and self.gen_token_idx[0] != -1:
          if i < min_word_tokens:
            logits[:, self.retrieval_token_idx] = filter_value
            logits[:, self.gen_token_idx] = filter_value
          else:
            if ret_scale_factor > 1:
              logits[:, self.retrieval_token_idx[0]] = logits[:, self.retrieval_token_idx[0]].abs() * ret_scale_factor
            if gen_scale_factor > 1:
              logits[:, self.gen_token_idx[0]] = logits[:, self.gen_token_idx[0]].abs() * gen_scale_factor
        if temperature == 0.0:
          if top_p != 1.0:
            raise ValueError('top_p cannot be set if temperature is 0 (greedy decoding).')
          next_token = torch.argmax(logits, keepdim=True, dim=-1)
        else:
          logits = logits / temperature
          if top_p < 1.0:
            assert top_p > 0, f'top_p should be above 0, got {top_p} instead.'
            sorted_logits, sorted_indices = torch.sort(logits, descending=True)
            cumulative_probs = torch.cumsum(torch.softmax(sorted_logits, dim=-1), dim=-1)
            remove = cumulative_probs > top_p
            remove[..., 1:] = remove[..., :-1].clone()
            remove[..., 0] = 0
            for j in range(sorted_indices.shape[0]):
              logits[j, sorted_indices[j, remove[j, :]]] = filter_value
          next_token = torch.multinomial(logits.exp(), 1)
        # Force generation of the remaining [IMG] tokens if [IMG0] is generated (batch 1 only, :518-520).
        if next_token.shape[0] == 1 and next_token.item() == self.retrieval_token_idx[0]:
          assert self.retrieval_token_idx == self.gen_token_idx, (self.retrieval_token_idx, self.gen_token_idx)
          next_token = torch.tensor(self.retrieval_token_idx)[None, :].long().to(dev)
        else:
          next_token = next_token.long().to(dev)
        out = next_token if out is None else torch.cat([out, next_token], dim=-1)
        next_embedding = self.input_embeddings(next_token)
        embeddings = torch.cat([embeddings, next_embedding.to(embeddings.dtype)], dim=1)
    return out, output_embeddings, output_logits


class GILL(nn.Module):
  def __init__(self, tokenizer, model_args: Optional[GILLArgs] = None, path_array: Optional[List[str]] = None,
               emb_matrix: Optional[torch.tensor] = None, load_sd: bool = False, num_gen_images: int = 1,
               decision_model_path: Optional[str] = None, sd_pipe=None):
    super().__init__()
    self.model = GILLModel(tokenizer, model_args)
    self.path_array = path_array
    self.emb_matrix = emb_matrix
    self.load_sd = load_sd
    self.num_gen_images = num_gen_images
    self.idx2dec = {0: 'gen', 1: 'ret', 2: 'same'}
    self.decision_model = None
    if load_sd:
      if sd_pipe is not None:          # synthetic / pre-built pipeline (tests, benchmarks)
        self.sd_pipe = sd_pipe
      else:                            # models.py:550-551
        from .sd import GillSDPipeline
        model_id = os.environ.get("GILL_SD_DIR", "runwayml/stable-diffusion-v1-5")
        self.sd_pipe = GillSDPipeline.from_pretrained(model_id).to("cuda")
    if decision_model_path is not None:
      # decision MLP of the gen-vs-ret choice (models.py:553-561); applied in generate_for_images_and_texts (:671-682)
      print('Loading decision model...')
      self.decision_model = nn.Sequential(*[nn.Dropout(0.5), nn.Linear(4096, 2)])
      mlp_checkpoint = torch.load(decision_model_path, map_location='cpu')
      self.decision_model.load_state_dict(mlp_checkpoint['state_dict'], strict=True)
      self.decision_model.eval()

  def __call__(self, images: Tensor, tgt_tokens: Optional[Tensor] = None, caption_len: Optional[Tensor] = None,
               generate: bool = False, num_words: int = 32, temperature: float = 1.0, top_p: float = 1.0,
               ret_scale_factor: float = 1.0, gen_scale_factor: float = 1.0, min_word_tokens: int = 0,
               mode: str = 'captioning', concat_captions: bool = False, input_prefix: Optional[str] = None) -> Tensor:
    if generate:
      return self.model.generate(images, num_words, temperature=temperature, top_p=top_p, min_word_tokens=min_word_tokens,
                                 ret_scale_factor=ret_scale_factor, gen_scale_factor=gen_scale_factor)
    return self.model(pixel_values=images, labels=tgt_tokens, caption_len=caption_len, mode=mode,
                      concat_captions=concat_captions, input_prefix=input_prefix)

  def generate_for_images_and_texts(self, prompts: List, num_words: int = 0, min_word_tokens: int = 0,
                                    ret_scale_factor: float = 1.0, gen_scale_factor: float = 1.0, top_p: float = 1.0,
                                    temperature: float = 0.0, max_num_rets: int = 1, generator=None,
                                    always_add_bos: bool = False, guidance_scale: float = 7.5, num_inference_steps: int = 50):
    """Encode prompts into embeddings, and generates text and image outputs accordingly (models.py:582-762)."""
    input_embs = []
    input_ids = []
    add_bos = True
    dev = self.model.logit_scale.device
    # argument errors of the reference (models.py:624, :629) are raised before any device work
    for p in prompts:
      if type(p) != str and not type(p).__module__.startswith('PIL'):
        raise ValueError(f'Input prompts should be either PIL.Image.Image or str types, got {type(p)} instead.')
    if num_words == 0:
      raise NotImplementedError('Generation not implemented for num_words=0.')
    with torch.no_grad():
      for p in prompts:
        if type(p) == str:
          text_ids = self.model.tokenizer(p, add_special_tokens=add_bos, return_tensors="pt").input_ids.to(dev)
          if not always_add_bos:
            add_bos = False
          input_embs.append(self.model.input_embeddings(text_ids))
          input_ids.append(text_ids)
        elif type(p).__module__.startswith('PIL'):
          # Encode as image (models.py:606-613)
          pixel_values = utils.get_pixel_values_for_model(self.model.feature_extractor, p)
          pixel_values = pixel_values.to(device=dev, dtype=self.model.logit_scale.dtype)[None, ...]
          visual_embs = self.model.get_visual_embs(pixel_values, mode='captioning')   # (1, n_visual_tokens, D)
          input_embs.append(visual_embs.to(input_embs[0].dtype) if input_embs else visual_embs)
        else:
          raise ValueError(f'Input prompts should be either PIL.Image.Image or str types, got {type(p)} instead.')
      input_embs = torch.cat(input_embs, dim=1)
      input_ids = torch.cat(input_ids, dim=1)

      if num_words == 0:
        raise NotImplementedError('Generation not implemented for num_words=0.')
      elif num_words > 0:
        generated_ids, generated_embeddings, _ = self.model.generate(
          input_embs, num_words, min_word_tokens=min_word_tokens, temperature=temperature, top_p=top_p,
          ret_scale_factor=ret_scale_factor, gen_scale_factor=gen_scale_factor)
        embeddings = generated_embeddings[-1][:, input_embs.shape[1]:]
        newline_token_id = self.model.tokenizer('\n', add_special_tokens=False).input_ids[0]
        trunc_idx = 0
        for j in range(generated_ids.shape[1]):
          if generated_ids[0, j] == newline_token_id:
            trunc_idx = j
            break
        if trunc_idx > 0:
          generated_ids = generated_ids[:, :trunc_idx]
          embeddings = embeddings[:, :trunc_idx]
      else:
        raise ValueError

      return_outputs = []
      all_ret_idx = [i for i, x in enumerate(generated_ids[0, :] == self.model.retrieval_token_idx[0]) if x][:max_num_rets]
      seen_image_idx = []  # Avoid showing the same image multiple times.
      last_ret_idx = 0
      if len(all_ret_idx) == 0:
        caption = self.model.tokenizer.batch_decode(generated_ids, skip_special_tokens=True)[0]
        return_outputs.append(utils.truncate_caption(caption))
      else:
        for ret_idx in all_ret_idx:
          assert generated_ids[0, ret_idx:ret_idx + self.model.num_tokens].cpu().detach().numpy().tolist() == self.model.retrieval_token_idx, (generated_ids[0, ret_idx:ret_idx + self.model.num_tokens], self.model.retrieval_token_idx)
          raw_emb = embeddings[:, ret_idx:ret_idx + self.model.num_tokens, :]  # (1, 8, 4096)
          assert len(self.model.args.text_emb_layers) == 1
          image_outputs = {'gen': [], 'ret': [], 'decision': None}
          if self.emb_matrix is not None:
            # Produce retrieval embedding (models.py:671-676); the Linear and the score GEMV run in libgill_amd
            from . import ops
            from PIL import UnidentifiedImageError
            ret_emb = self.model.ret_text_hidden_fcs[0](raw_emb, None)[:, 0, :]  # (1, 256)
            ret_emb = ret_emb / ret_emb.norm(dim=-1, keepdim=True)
            ret_emb = ret_emb.type(self.emb_matrix.dtype)  # (1, 256)
            scores = self._scores(self.emb_matrix, ret_emb)                         # emb_matrix @ ret_emb.T, (N, 1)
            for seen_idx in seen_image_idx:   # Downweight seen images.
              scores[seen_idx, :] -= 1000
            _, top_image_idx = scores.squeeze().topk(3)
            for img_idx in top_image_idx:     # Find the first image that does not error out.
              try:
                seen_image_idx.append(img_idx)
                img = utils.get_image_from_url(self.path_array[img_idx])
                image_outputs['ret'].append((img, 'ret', scores[img_idx].item()))
                if len(image_outputs) == max_num_rets:
                  break
              except (UnidentifiedImageError, ConnectionError, OSError):
                pass
            if self.decision_model is not None:   # Make decision with MLP (models.py:698-704)
              decision_emb = raw_emb[:, 0, :]  # (1, 4096)
              lin = self.decision_model[1]
              assert decision_emb.shape[1] == lin.in_features, decision_emb.shape
              w4 = torch.zeros((4, lin.in_features), device=decision_emb.device, dtype=torch.bfloat16)
              w4[:lin.out_features] = lin.weight.to(decision_emb.device, torch.bfloat16)
              b4 = torch.zeros((4,), device=decision_emb.device)
              b4[:lin.out_features] = lin.bias.to(decision_emb.device).float()
              decision_logits = ops.gemm(decision_emb, w4, bias=b4, out_f32=True)[:, :lin.out_features]
              probs = decision_logits.softmax(dim=-1).cpu().float().numpy().tolist()
              image_outputs['decision'] = [self.idx2dec[decision_logits.argmax().item()]] + probs
          else:
            # If no embedding matrix is provided, generate instead.
            image_outputs['decision'] = ['gen', [0, 1]]

          gen_prefix = ''.join([f'[IMG{i}]' for i in range(self.model.args.num_tokens)])
          gen_prefx_ids = self.model.tokenizer(gen_prefix, add_special_tokens=False, return_tensors="pt").input_ids.to(dev)
          gen_prefix_embs = self.model.input_embeddings(gen_prefx_ids)  # (1, T, D)
          gen_emb = self.model.gen_text_hidden_fcs[0](raw_emb, gen_prefix_embs)  # (1, 77, 768)
          if gen_emb.shape[1] != 77:
            print(f"Padding {gen_emb.shape} with zeros")
            bs = gen_emb.shape[0]
            clip_emb = 768
            gen_emb = gen_emb.reshape(bs, -1, clip_emb)
            seq_len = gen_emb.shape[1]
            gen_emb = torch.cat([gen_emb, torch.zeros((bs, 77 - seq_len, clip_emb), device=gen_emb.device, dtype=gen_emb.dtype)], dim=1)
            print('Padded to', gen_emb.shape)
          gen_emb = gen_emb.repeat(self.num_gen_images, 1, 1)  # (self.num_gen_images, 77, 768)

          if self.load_sd:
            gen_max_bs = 8
            gen_images = []
            for i in range(0, self.num_gen_images, gen_max_bs):
              gen_images.extend(
                self.sd_pipe(prompt_embeds=gen_emb[i:i + gen_max_bs], generator=generator, guidance_scale=guidance_scale,
                             num_inference_steps=num_inference_steps,
                             output_type="pil" if getattr(self.sd_pipe, "_vae", None) else "latent").images)
            # PIL images when the pipeline holds VAE weights (reference behaviour), else the final latents (4,64,64)
            if self.emb_matrix is not None and getattr(self.sd_pipe, "_vae", None):
              # CLIP rerank of the generated images against the retrieval embedding (models.py:733-751)
              all_gen_pixels = []
              for img in gen_images:
                pixel_values = utils.get_pixel_values_for_model(self.model.feature_extractor, img.resize((224, 224)).convert('RGB'))
                all_gen_pixels.append(pixel_values.to(device=dev, dtype=self.model.logit_scale.dtype))
              all_gen_pixels = torch.stack(all_gen_pixels, dim=0)
              gen_visual_embs = self.model.get_visual_embs(all_gen_pixels, mode='retrieval')  # (n, 1, D)
              gen_visual_embs = gen_visual_embs / gen_visual_embs.norm(dim=-1, keepdim=True)
              gen_visual_embs = gen_visual_embs.type(self.emb_matrix.dtype)
              gen_rank_scores = self._scores(gen_visual_embs.reshape(len(gen_images), -1), ret_emb).squeeze()
              sorted_score_idx = torch.argsort(-gen_rank_scores)
              if self.num_gen_images > 1:   # Rank images by retrieval score.
                image_outputs['gen'] = [(gen_images[idx], gen_rank_scores[idx].item()) for idx in sorted_score_idx]
              else:
                image_outputs['gen'] = [(gen_images[0], gen_rank_scores.item())]
            else:
              image_outputs['gen'] = [(gen_images[0], 0)]
          else:
            image_outputs['gen'] = [gen_emb]

          caption = self.model.tokenizer.batch_decode(generated_ids[:, last_ret_idx:ret_idx], skip_special_tokens=True)[0]
          last_ret_idx = ret_idx + 1
          return_outputs.append(utils.truncate_caption(caption) + f' {gen_prefix}')
          return_outputs.append(image_outputs)
    return return_outputs

  @torch.no_grad()
  def get_log_likelihood_scores(self, prompts: List) -> float:
    """Log likelihood of an interleaved image / text prompt: minus the LM's mean next-token cross-entropy over the TEXT positions
    (image positions carry the label -100 and are skipped, the <bos> tag is added once) — models.py:764-807, whose
    `self.model.lm(inputs_embeds=..., labels=...)` is the OPT forward plus the tied lm_head and HF's shifted CrossEntropyLoss.
    Here: hidden_states[-1] from gill_opt_forward, the lm_head rows of the T - 1 predicting positions through
    gill_opt_last_logits (8 positions per call), and the log-softmax / mean on the (T - 1, vocab) fp32 host tensor."""
    dev = self.model.logit_scale.device
    embs, ids, first_text = [], [], True
    for p in prompts:
      if type(p) == str:
        t = self.model.tokenizer(p, add_special_tokens=True, return_tensors="pt").input_ids.to(dev)
        if not first_text:
          t = t[:, 1:]          # <bos> only once
        first_text = False
        embs.append(self.model.input_embeddings(t))
        ids.append(t)
      elif type(p).__module__.startswith('PIL'):
        px = utils.get_pixel_values_for_model(self.model.feature_extractor, p)
        px = px.to(device=dev, dtype=self.model.logit_scale.dtype)[None, ...]
        v = self.model.get_visual_embs(px, mode='captioning')
        embs.append(v)
        ids.append(torch.full(v.shape[:2], -100, dtype=torch.int64, device=dev))
      else:
        raise ValueError(f'Input prompts should be either PIL.Image.Image or str types, got {type(p)} instead.')
    dt = next(e.dtype for e in embs)
    x = torch.cat([e.to(dt) for e in embs], dim=1)
    labels = torch.cat(ids, dim=1)[0, 1:].cpu()                  # position t predicts token t + 1
    hidden = self.model._lm_forward_hidden(x)                      # (1, T, D) fp32, post final LayerNorm
    T, D = hidden.shape[1], hidden.shape[2]
    if T < 2 or bool((labels == -100).all()):
      return float('nan')                                          # HF's mean over zero targets
    vocab = self.model.opt_cfg.vocab_size
    rows = hidden[0, :T - 1].contiguous()
    logits = torch.empty((T - 1, vocab), device=dev, dtype=torch.float32)
    with torch.cuda.device(dev):
      for r0 in range(0, T - 1, 8):
        nb = min(8, T - 1 - r0)
        # (nb, 1, D): every row is the "last position" of its own one-token sequence
        N.check(N.lib().gill_opt_last_logits(self.model._opt_handle, N.ptr(rows[r0:r0 + nb]), nb, 1, N.ptr(logits[r0:r0 + nb]),
                                             N.current_stream()))
    loss = F.cross_entropy(logits.cpu(), labels, ignore_index=-100)
    return -loss.item()

  @staticmethod
  def _scores(matrix: Tensor, query: Tensor) -> Tensor:
    """matrix (N, D) @ query (1, D).T -> (N, 1) in matrix's dtype, through the native GEMM (the query is padded to the 4
    output columns the kernel's epilogue writes at a time)."""
    from . import ops
    dev = query.device
    Nq, D = matrix.shape
    Dp = (D + 63) // 64 * 64
    a = torch.zeros((Nq, Dp), device=dev, dtype=torch.bfloat16)
    a[:, :D] = matrix.to(dev)
    q4 = torch.zeros((4, Dp), device=dev, dtype=torch.bfloat16)
    q4[0, :D] = query.reshape(-1).to(torch.bfloat16)
    return ops.gemm(a, q4, out_f32=True)[:, :1].to(matrix.dtype)

  # ---- NEW, build-defined batched entry (not a reference function) ---------------------------------------
  @torch.no_grad()
  def generate_images(self, prompts, num_inference_steps: int = 50, guidance_scale: float = 7.5,
                      latents: Optional[Tensor] = None, seed: int = 1337, return_latents: bool = True,
                      return_embeddings: bool = False, distributed: bool = True, decode: bool = False):
    """Batched text -> image latents.  `prompts` is a list of strings (tokenized here) or an int64 tensor
    (B,T) of prompt token ids WITHOUT the [IMG] tokens (right-padded with pad_token_id).  Per prompt this equals the
    'gen' branch of generate_for_images_and_texts([p], num_words=2, gen_scale_factor=1e5): the 8 [IMG] ids are
    appended, one OPT pass yields their hidden states (models.py:384), the GILLMapper maps them to the (77,768)
    SD conditioning (models.py:387/710) and the SD-1.5 UNet runs the CFG/PLMS loop (custom_sd.py:607-651).
    With torch.distributed initialised (one process per GPU, RCCL) the prompts are sharded contiguously over
    ranks and the final latents are all-gathered (the only collective of the path).  decode=True additionally runs
    the VAE decoder (custom_sd.py:385-392) on THIS rank's shard and returns (latents_all, images_local) with
    images_local (B_local,512,512,3) uint8 on the device; decoded pixels are never exchanged between ranks."""
    from . import parallel
    dev = self.model.logit_scale.device
    if (isinstance(prompts, torch.Tensor) and prompts.numel() == 0) or (not isinstance(prompts, torch.Tensor) and len(prompts) == 0):
      raise ValueError('generate_images: empty prompt batch')
    ids, lens = self._prompt_ids(prompts)
    B_total = ids.shape[0]
    lo, hi = parallel.shard_range(B_total, distributed)
    ids, lens = ids[lo:hi], lens[lo:hi]
    B = ids.shape[0]
    img = torch.tensor(self.model.gen_token_idx, dtype=torch.int64)
    k = self.model.num_tokens
    T = int(lens.max().item()) + k if B > 0 else k
    pad = self.model.tokenizer.pad_token_id
    full = torch.full((B, T), pad if pad is not None else 1, dtype=torch.int64)
    for b in range(B):
      n = int(lens[b])
      full[b, :n] = ids[b, :n]
      full[b, n:n + k] = img
    last_idx = lens + k - 1
    # a rank whose shard is empty (fewer prompts than ranks) still enters the collectives below with (0, ...) tensors
    embs = torch.zeros((0, self.model.args.num_clip_tokens, self.model.args.gen_emb_dim), device=dev,
                       dtype=self.model.logit_scale.dtype)
    local = None
    if self.load_sd:
      local = torch.zeros((0, self.sd_pipe.cfg.in_channels, self.sd_pipe.cfg.sample_size, self.sd_pipe.cfg.sample_size),
                          device=dev, dtype=torch.float32)
    if B > 0:
      raw, emb = self.model.img_hidden_states(full.to(dev), last_idx)
      embs = self.model.gen_text_hidden_fcs[0](raw, emb)          # (B,77,768)
      if self.load_sd:
        lat0 = None
        if latents is not None:
          lat0 = latents[lo:hi]
        else:
          from .synth import initial_latents
          lat0 = initial_latents(B_total, self.sd_pipe.cfg.in_channels, self.sd_pipe.cfg.sample_size, seed)[lo:hi]
        outs = []
        for i in range(0, B, 8):                                    # gen_max_bs = 8 (models.py:726)
          outs.append(self.sd_pipe(prompt_embeds=embs[i:i + 8], latents=lat0[i:i + 8], guidance_scale=guidance_scale,
                                   num_inference_steps=num_inference_steps, output_type="latent").images)
        local = torch.cat(outs, 0)
    images = None
    if decode:
      if not self.load_sd:
        raise ValueError('decode=True needs load_sd=True')
      images = self.sd_pipe.decode_latents(local, as_uint8=True) if B > 0 else \
          torch.zeros((0, 8 * self.sd_pipe.cfg.sample_size, 8 * self.sd_pipe.cfg.sample_size, 3), device=dev, dtype=torch.uint8)
    if not self.load_sd:
      return parallel.gather_rows(embs, B_total, distributed)
    out = parallel.gather_rows(local, B_total, distributed)
    if decode:
      return out, images
    if return_embeddings:
      return out, parallel.gather_rows(embs.float(), B_total, distributed)
    return out

  def _prompt_ids(self, prompts):
    if isinstance(prompts, torch.Tensor):
      ids = prompts.to('cpu', torch.int64)
      pad = self.model.tokenizer.pad_token_id
      # length = row length minus the TRAILING run of pad ids (a global count of non-pad ids miscounts when pad == eos/bos,
      # which load_gill sets for tokenizers without a pad token)
      if pad is None:
        lens = torch.full((ids.shape[0],), ids.shape[1])
      else:
        not_pad = (ids != pad)
        last = torch.where(not_pad.any(dim=1), ids.shape[1] - 1 - torch.argmax(not_pad.flip(1).to(torch.int64), dim=1),
                           torch.full((ids.shape[0],), -1))
        lens = last + 1
      return ids, lens.to(torch.int64)
    rows = [self.model.tokenizer(p, add_special_tokens=True, return_tensors="pt").input_ids[0] for p in prompts]
    lens = torch.tensor([len(r) for r in rows], dtype=torch.int64)
    T = int(lens.max())
    pad = self.model.tokenizer.pad_token_id
    ids = torch.full((len(rows), T), pad if pad is not None else 1, dtype=torch.int64)
    for i, r in enumerate(rows):
      ids[i, :len(r)] = r
    return ids, lens


def load_gill(model_dir: str, load_ret_embs: bool = True, decision_model_fn: str = 'decision_model.pth.tar') -> GILL:
  """reference: gill/models.py:810-902.  Same files, same errors, same tokenizer surgery, same checkpoint format."""
  model_args_path = os.path.join(model_dir, 'model_args.json')
  model_ckpt_path = os.path.join(model_dir, 'pretrained_ckpt.pth.tar')
  embs_paths = [s for s in glob.glob(os.path.join(model_dir, 'cc3m*.npy'))]
  if not os.path.exists(model_args_path):
    raise ValueError(f'model_args.json does not exist in {model_dir}.')
  if not os.path.exists(model_ckpt_path):
    raise ValueError(f'pretrained_ckpt.pth.tar does not exist in {model_dir}.')
  path_array, emb_matrix = None, None
  if load_ret_embs and embs_paths:
    # models.py:826-838: every cc3m*.npy is a pickle {'paths': [...], 'embeddings': [...]} precomputed with
    # get_visual_embs(image, mode='retrieval'); rows of all files are concatenated in glob order
    path_array, emb_matrix = _read_retrieval_embeddings(embs_paths)
  else:
    if not embs_paths:
      print(f'cc3m.npy files do not exist in {model_dir}.')
    print('Running the model without retrieval.')
  with open(model_args_path, 'r') as f:
    model_kwargs = json.load(f)

  from transformers import AutoTokenizer
  tokenizer = AutoTokenizer.from_pretrained(model_kwargs['opt_version'], use_fast=False)
  if tokenizer.pad_token is None:
    tokenizer.pad_token_id = tokenizer.eos_token_id
  tokenizer.add_special_tokens({"cls_token": "<|image|>"})
  model_kwargs['retrieval_token_idx'] = []
  for i in range(model_kwargs['num_tokens']):
    print(f'Adding [IMG{i}] token to vocabulary.')
    tokenizer.add_tokens(f'[IMG{i}]')
    ret_token_idx = tokenizer(f'[IMG{i}]', add_special_tokens=False).input_ids
    assert len(ret_token_idx) == 1, ret_token_idx
    model_kwargs['retrieval_token_idx'].append(ret_token_idx[0])
  model_kwargs['gen_token_idx'] = model_kwargs['retrieval_token_idx']
  args = namedtuple('args', model_kwargs)(**model_kwargs)

  decision_model_path = os.path.join(model_dir, decision_model_fn) if decision_model_fn is not None else None
  model = GILL(tokenizer, args, path_array=path_array, emb_matrix=emb_matrix, load_sd=True, num_gen_images=1,
               decision_model_path=decision_model_path)
  model = model.eval()
  model = model.bfloat16()
  model = model.cuda()

  checkpoint = torch.load(model_ckpt_path, map_location='cpu')
  state_dict = {}
  for k, v in checkpoint['state_dict'].items():
    state_dict[k.replace('module.', '')] = v
  img_token_embeddings = state_dict['model.input_embeddings.weight'].cpu().detach()
  del state_dict['model.input_embeddings.weight']
  model.load_state_dict(state_dict, strict=False)
  with torch.no_grad():
    if 'share_ret_gen' in model_kwargs:
      assert model_kwargs['share_ret_gen'], 'Model loading only supports share_ret_gen=True for now.'
    model.model.input_embeddings.weight[-model_kwargs['num_tokens']:, :].copy_(img_token_embeddings)
  model.model.refresh_native()   # native handles snapshot the weights: rebuild after the in-place copy
  if emb_matrix is not None:     # models.py:895-900: unit rows scaled by exp(logit_scale), in the model's dtype, on its device
    scale = model.model.logit_scale.exp()
    m = torch.as_tensor(emb_matrix).to(device=scale.device, dtype=scale.dtype)
    model.emb_matrix = scale * (m / m.norm(dim=1, keepdim=True))
  return model


def _read_retrieval_embeddings(paths: List[str]):
  """cc3m*.npy files (pickles despite the suffix) -> (list of image paths / urls, float array (N, ret_emb_dim))."""
  import pickle
  names: List[str] = []
  rows = []
  for p in paths:
    with open(p, 'rb') as f:
      blob = pickle.load(f)
    names.extend(blob['paths'])
    rows.extend(blob['embeddings'])
  mat = np.stack(rows, axis=0)
  assert len(names) == mat.shape[0], (len(names), mat.shape)   # one embedding per path (models.py:841)
  return names, mat
