"""Mirror of gill/layers.py (reference) — `TextFcLayer`, whose 'gill_mapper' mode is the GILLMapper.

Same constructor, same parameter / state-dict names (fc.*, tfm.encoder.layers.N.*, tfm.decoder.layers.N.*,
model.*, query_embs), same forward signature and assertions; the arithmetic of forward() runs in
libgill_amd (gill_mapper_forward: csrc/mapper.hip) instead of torch.nn.Transformer.
nn.Transformer / nn.Linear are instantiated ONLY as parameter containers so that checkpoints written by the
reference (scripts/prune_model_ckpt.py) load unchanged; their forward() is never called.
"""
from __future__ import annotations

import ctypes as C

import torch
from torch import nn

from . import _native as N


class TextFcLayer(nn.Module):
  """Layers used in mapping text embeddings to visual outputs.  (reference: gill/layers.py:5-53)"""

  def __init__(self, in_dim: int, out_dim: int, num_input_tokens: int = 1, num_output_tokens: int = 1, mode: str = 'linear'):
    super().__init__()
    self.num_input_tokens = num_input_tokens
    self.num_output_tokens = num_output_tokens
    self.mode = mode
    self.in_dim, self.out_dim = in_dim, out_dim
    if mode == 'linear':
      self.model = nn.Linear(in_dim, out_dim)
    elif mode == 'gill_mapper':
      hidden_dim = 512
      self.hidden_dim = hidden_dim
      self.fc = nn.Linear(in_dim, hidden_dim)
      self.tfm = nn.Transformer(batch_first=True, norm_first=True, d_model=hidden_dim, num_encoder_layers=4,
                                num_decoder_layers=4, dim_feedforward=hidden_dim * 4, dropout=0.0, nhead=4)
      self.model = nn.Linear(hidden_dim, out_dim)
      self.query_embs = nn.Parameter(torch.randn(1, num_output_tokens, hidden_dim))
    else:
      raise NotImplementedError(mode)
    self._handle = None
    self._handle_key = None
    # the native handle snapshots the weights: any state-dict load (also the recursive one of a parent module, which does not
    # go through this class's load_state_dict) drops it
    self.register_load_state_dict_post_hook(lambda module, incompatible_keys: module.release_native())

  # ---- native handle management -------------------------------------------------------------
  def _apply(self, fn, *a, **k):          # .cuda() / .bfloat16() / .to(): weights move -> rebuild lazily
    self.release_native()
    return super()._apply(fn, *a, **k)

  def load_state_dict(self, *a, **k):
    self.release_native()
    return super().load_state_dict(*a, **k)

  def release_native(self):
    if getattr(self, "_handle", None):
      N.lib().gill_mapper_destroy(self._handle)
    self._handle = None
    self._handle_key = None

  def __del__(self):
    try:
      self.release_native()
    except Exception:
      pass

  def _native(self, batch: int):
    dev = self.query_embs.device
    if dev.type != "cuda":
      raise N.GillNativeError("TextFcLayer('gill_mapper') runs only on an MI355X through libgill_amd; "
                              "move the module to cuda (there is no CPU implementation in this package).")
    cap = max(8, batch)
    key = (dev.index, cap)
    if self._handle is not None and self._handle_key[0] == dev.index and self._handle_key[1] >= batch:
      return self._handle
    self.release_native()
    cfg = N.gill_mapper_config(in_dim=self.in_dim, out_dim=self.out_dim, hidden_dim=self.hidden_dim, num_heads=4,
                               ffn_dim=self.hidden_dim * 4, num_enc_layers=4, num_dec_layers=4,
                               num_input_tokens=self.num_input_tokens, num_output_tokens=self.num_output_tokens,
                               max_batch=cap)
    sd = {k: v for k, v in self.state_dict().items()}
    arr, keep = N.make_tensor_table(sd, dev)
    h = C.c_void_p()
    with torch.cuda.device(dev):
      N.check(N.lib().gill_mapper_create(C.byref(h), C.byref(cfg), arr, len(sd)))
    del keep
    self._handle, self._handle_key = h, key
    return h

  # ---- forward ------------------------------------------------------------------------------
  def forward(self, x: torch.Tensor, input_embs: torch.Tensor) -> torch.Tensor:
    outputs = None
    if self.mode == 'gill_mapper':
      B = x.shape[0]
      assert x.shape[1] == self.num_input_tokens and x.shape[2] == self.in_dim, x.shape
      h = self._native(B)
      xb = x.to(torch.bfloat16).contiguous()
      eb = None
      Be = 0
      if input_embs is not None:
        eb = input_embs.to(torch.bfloat16).contiguous()
        Be = eb.shape[0]
        assert Be in (1, B), (eb.shape, x.shape)
      out = torch.empty((B, self.num_output_tokens, self.out_dim), device=x.device, dtype=torch.float32)
      with torch.cuda.device(x.device):
        N.check(N.lib().gill_mapper_forward(h, N.ptr(xb), N.ptr(eb), B, Be, N.ptr(out), N.current_stream()))
      outputs = out.to(x.dtype) if x.dtype != torch.float32 else out
    elif self.mode == 'linear':
      # ret_text_fc_mode: one Linear over every token (layers.py:44-49), then the first num_output_tokens tokens
      from . import ops
      Bn, T, D = x.shape
      y = ops.gemm(x.reshape(Bn * T, D), self.model.weight, bias=self.model.bias, out_f32=True).reshape(Bn, T, self.out_dim)
      if y.shape[1] != self.num_output_tokens:
        y = y[:, :self.num_output_tokens, :]
      outputs = y.to(x.dtype) if x.dtype != torch.float32 else y
    else:
      raise NotImplementedError(f"TextFcLayer mode {self.mode!r}")
    # layers.py:52 hard-codes the SD-1.x width 768; a mapper built with another out_dim (gen_emb_dim = 1024: the SD-2.x variant
    # main.py:253 anticipates, BASELINE configs[3]) is checked against its own width
    width = 768 if self.out_dim in (256, 768) else self.out_dim
    assert outputs.shape[1] == 1 or (outputs.shape[1] * outputs.shape[2] == self.num_output_tokens * width), (outputs.shape, self.num_output_tokens)
    return outputs  # (N, T, D)
