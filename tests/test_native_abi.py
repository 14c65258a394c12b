"""CPU-side checks of the drop-in boundary: libgill_amd.so loads, exports every symbol include/gill_amd.h declares
(and nothing in the ctypes table is missing from the header), host-only entry points answer without a GPU, and the
product path fails loudly — never silently falls back — when no GPU is present."""
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
  src = open(os.path.join(ROOT, "include", "gill_amd.h")).read()
  src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
  return sorted(set(re.findall(r"\b(gill_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_header_symbol():
  from gill_amd import _native as N
  lib = N.lib()
  names = _header_functions()
  assert len(names) >= 20
  for n in names:
    assert hasattr(lib, n), f"{n} declared in include/gill_amd.h but not exported by libgill_amd.so"
  assert sorted(N.SYMBOLS) == names, "ctypes table and header disagree"
  assert lib.gill_version() >= 100


def test_pndm_schedule_host_entry_matches_oracle():
  import ctypes as C
  from gill_amd import _native as N
  from oracle.scheduler_ref import PNDMSchedulerRef
  for steps in (10, 50):
    ts = (C.c_int32 * (steps + 1))()
    ac = (C.c_double * 1000)()
    n = N.lib().gill_pndm_schedule(steps, ts, ac)
    ref = PNDMSchedulerRef()
    assert list(ts)[:n] == ref.set_timesteps(steps) and n == steps + 1
    err = max(abs(ac[i] - float(ref.alphas_cumprod[i])) / float(ref.alphas_cumprod[i]) for i in range(1000))
    assert err < 5e-6, err


def test_bad_arguments_report_errors_not_crashes():
  import ctypes as C
  from gill_amd import _native as N
  lib = N.lib()
  assert lib.gill_pndm_schedule(1, None, None) != 0
  assert b"num_steps" in lib.gill_last_error()
  h = C.c_void_p()
  assert lib.gill_mapper_create(C.byref(h), None, None, 0) != 0
  assert lib.gill_last_error()


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_product_path_fails_loudly_without_gpu():
  from gill_amd import _native as N
  from gill_amd.layers import TextFcLayer
  layer = TextFcLayer(768, 768, num_input_tokens=8, num_output_tokens=77, mode="gill_mapper")
  with pytest.raises(N.GillNativeError):
    layer(torch.zeros(1, 8, 768), torch.zeros(1, 8, 768))


def test_state_dict_names_match_reference_layout():
  """TextFcLayer keeps the reference's parameter names (gill/layers.py:17-24) so its checkpoints load unchanged."""
  from gill_amd import synth
  from gill_amd.layers import TextFcLayer
  layer = TextFcLayer(768, 768, num_input_tokens=8, num_output_tokens=77, mode="gill_mapper")
  sd = synth.mapper_state_dict(synth.MapperConfig(in_dim=768), seed=0)
  assert sorted(layer.state_dict().keys()) == sorted(sd.keys())
  layer.load_state_dict(sd, strict=True)
