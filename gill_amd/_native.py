"""ctypes binding of libgill_amd.so (C ABI declared in include/gill_amd.h).

The library is the product: there is no Python/torch fallback for any op it exports.  Importing this
module without a built library raises; calling into it without a GPU fails in the HIP runtime.
torch is used only as the owner of device memory and streams (tensor.data_ptr(), current stream).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GILL_AMD_LIB") or os.path.join(_HERE, "libgill_amd.so")   # GILL_AMD_LIB: A/B runs of two builds


class GillNativeError(RuntimeError):
  pass


class gill_tensor(C.Structure):
  _fields_ = [("name", C.c_char_p), ("data", C.c_void_p), ("dtype", C.c_int32), ("ndim", C.c_int32),
              ("shape", C.c_int64 * 4)]


class gill_opt_config(C.Structure):
  _fields_ = [(n, C.c_int32) for n in ("vocab_size", "hidden_size", "num_layers", "num_heads", "ffn_dim",
                                       "max_positions", "max_batch", "max_seq")]


class gill_mapper_config(C.Structure):
  _fields_ = [(n, C.c_int32) for n in ("in_dim", "out_dim", "hidden_dim", "num_heads", "ffn_dim", "num_enc_layers",
                                       "num_dec_layers", "num_input_tokens", "num_output_tokens", "max_batch")]


class gill_unet_config(C.Structure):
  _fields_ = [("in_channels", C.c_int32), ("out_channels", C.c_int32), ("block_out_channels", C.c_int32 * 4),
              ("layers_per_block", C.c_int32), ("cross_attention_dim", C.c_int32), ("num_heads", C.c_int32),
              ("norm_num_groups", C.c_int32), ("sample_size", C.c_int32), ("ctx_len", C.c_int32),
              ("max_batch", C.c_int32), ("heads_per_level", C.c_int32 * 4), ("v_prediction", C.c_int32),
              ("fp8_convs", C.c_int32)]


class gill_clip_config(C.Structure):
  _fields_ = [(n, C.c_int32) for n in ("image_size", "patch_size", "hidden_size", "num_layers", "num_heads", "intermediate_size",
                                       "max_batch")]


class gill_vae_config(C.Structure):
  _fields_ = [("latent_channels", C.c_int32), ("out_channels", C.c_int32), ("block_out_channels", C.c_int32 * 4),
              ("layers_per_block", C.c_int32), ("norm_num_groups", C.c_int32), ("latent_size", C.c_int32),
              ("scaling_factor", C.c_float), ("max_batch", C.c_int32)]


# every symbol include/gill_amd.h declares: (restype, argtypes)
_vp, _i, _f = C.c_void_p, C.c_int, C.c_float
SYMBOLS: Dict[str, Tuple[object, List[object]]] = {
  "gill_last_error": (C.c_char_p, []),
  "gill_version": (_i, []),
  "gill_opt_create": (_i, [C.POINTER(_vp), C.POINTER(gill_opt_config), C.POINTER(gill_tensor), _i]),
  "gill_opt_destroy": (None, [_vp]),
  "gill_opt_embed": (_i, [_vp, _vp, _i, _vp, _vp]),
  "gill_opt_forward": (_i, [_vp, _vp, _i, _i, _vp, _vp]),
  "gill_opt_forward_cached": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
  "gill_opt_img_hidden": (_i, [_vp, _vp, C.POINTER(C.c_int32), _i, _i, _i, _vp, _vp, _vp]),
  "gill_opt_last_logits": (_i, [_vp, _vp, _i, _i, _vp, _vp]),
  "gill_clip_create": (_i, [C.POINTER(_vp), C.POINTER(gill_clip_config), C.POINTER(gill_tensor), _i]),
  "gill_clip_destroy": (None, [_vp]),
  "gill_clip_forward": (_i, [_vp, _vp, _i, _vp, _vp]),
  "gill_mapper_create": (_i, [C.POINTER(_vp), C.POINTER(gill_mapper_config), C.POINTER(gill_tensor), _i]),
  "gill_mapper_destroy": (None, [_vp]),
  "gill_mapper_forward": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp]),
  "gill_unet_create": (_i, [C.POINTER(_vp), C.POINTER(gill_unet_config), C.POINTER(gill_tensor), _i]),
  "gill_unet_destroy": (None, [_vp]),
  "gill_unet_forward": (_i, [_vp, _vp, C.POINTER(C.c_float), _vp, _i, _vp, _vp]),
  "gill_sd_denoise": (_i, [_vp, _vp, _vp, _i, _vp, _i, _i, _f, _vp, _vp]),
  "gill_vae_create": (_i, [C.POINTER(_vp), C.POINTER(gill_vae_config), C.POINTER(gill_tensor), _i]),
  "gill_vae_destroy": (None, [_vp]),
  "gill_vae_decode": (_i, [_vp, _vp, _i, _vp, _vp, _vp]),
  "gill_op_conv3x3_fp8": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
  "gill_pndm_schedule": (_i, [_i, C.POINTER(C.c_int32), C.POINTER(C.c_double)]),
  "gill_coop_timeouts": (_i, []),
  "gill_op_gemm": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _i, _i, _i, _vp]),
  "gill_op_geglu": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
  "gill_op_conv3x3": (_i, [_vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
  "gill_op_attention": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _i, _vp]),
  "gill_op_layernorm": (_i, [_vp, _i, _vp, _vp, _vp, _i, _i, _f, _vp]),
  "gill_op_groupnorm": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _vp, _vp, _f, _i, _vp, _vp]),
  "gill_op_conv3x3_gn": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _f, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
  "gill_op_conv3x3_shortcut": (_i, [_vp, _i, _vp, _i, _vp, _vp, _vp, _i, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
  "gill_op_ffn_fused": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp]),
  "gill_op_geglu_fp8": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
  "gill_op_lnproj": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp]),
  "gill_op_cross_attention_folded": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
}

_lib: Optional[C.CDLL] = None


def lib() -> C.CDLL:
  """Load libgill_amd.so (once).  Raises GillNativeError when it has not been built."""
  global _lib
  if _lib is None:
    if not os.path.exists(LIB_PATH):
      raise GillNativeError(
        f"{LIB_PATH} is missing: the HIP extension is the only implementation of the GILL hot path. "
        "Build it with `python -c 'import __graft_entry__ as g; g.build()'` (or `make -C gill_amd/csrc`).")
    l = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
      fn = getattr(l, name)  # AttributeError here == header/library mismatch
      fn.restype = res
      fn.argtypes = args
    _lib = l
  return _lib


def check(rc: int) -> None:
  if rc != 0:
    msg = lib().gill_last_error()
    raise GillNativeError(f"libgill_amd error {rc}: {msg.decode() if msg else '?'}")


def current_stream() -> int:
  return int(torch.cuda.current_stream().cuda_stream)


_DTYPES = {torch.bfloat16: 0, torch.float32: 1, torch.float16: 2}


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
  if t is None:
    return None
  assert t.is_cuda and t.is_contiguous(), "libgill_amd takes contiguous device tensors"
  return t.data_ptr()


def make_tensor_table(state: Dict[str, torch.Tensor], device: torch.device):
  """state-dict -> (gill_tensor array, keep-alive list).  Tensors are moved to `device` if needed."""
  keep = []
  arr = (gill_tensor * len(state))()
  for i, (k, v) in enumerate(state.items()):
    if v.dtype not in _DTYPES:
      v = v.float()
    v = v.detach().to(device).contiguous()
    name = k.encode()
    keep.append((name, v))
    arr[i].name = name
    arr[i].data = v.data_ptr()
    arr[i].dtype = _DTYPES[v.dtype]
    arr[i].ndim = v.dim()
    for d in range(4):
      arr[i].shape[d] = v.shape[d] if d < v.dim() else 1
  return arr, keep
