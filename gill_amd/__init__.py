"""gill_amd — MI355X-native implementation of the GILL image-generation hot path
(OPT [IMG] hidden states -> GILLMapper -> SD-1.5 UNet denoise loop) behind the reference's
gill.models / gill.layers Python surface.  All compute runs in libgill_amd.so (hand-written HIP)."""
__version__ = "0.1.0"


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
