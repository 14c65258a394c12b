"""gill_amd — MI355X-native implementation of the GILL image-generation hot path
(OPT [IMG] hidden states -> GILLMapper -> SD-1.5 UNet denoise loop) behind the reference's
gill.models / gill.layers Python surface.  All compute runs in libgill_amd.so (hand-written HIP)."""
import os as _os

__version__ = "0.1.0"

# ROCm 7.2's hipGraph "packet capture" costs the captured denoise loop 1.2 % (469.0 vs 463.4 ms per 4-prompt loop: profiles/
# r04_weight_prefetch.md) and was the mechanism behind round 1's mis-replayed memset nodes (profiles/r02_soak_bisect.md).  The runtime
# reads the flag once, when HIP initialises: importing gill_amd before the first CUDA call makes it effective; later it is a no-op.
_os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")


def install_as_gill() -> None:
  """Register this package under the reference's import name, so that code written against kohjingyu/gill
  (`from gill import models`, `import gill.layers`, `gill.utils`) runs unchanged on the MI355X path:

      import gill_amd; gill_amd.install_as_gill()
      from gill import models            # gill_amd.models
      model = models.load_gill(model_dir)

  Refuses to shadow a real `gill` package that is already imported."""
  import importlib
  import sys
  me = sys.modules[__name__]
  other = sys.modules.get("gill")
  if other is not None and other is not me:
    raise ImportError("a different `gill` package is already imported; install_as_gill() must run before `import gill`")
  sys.modules["gill"] = me
  for sub in ("models", "layers", "utils"):
    sys.modules["gill." + sub] = importlib.import_module(__name__ + "." + sub)
