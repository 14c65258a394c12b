"""gill_amd — MI355X-native implementation of the GILL image-generation hot path
(OPT [IMG] hidden states -> GILLMapper -> SD-1.5 UNet denoise loop) behind the reference's
gill.models / gill.layers Python surface.  All compute runs in libgill_amd.so (hand-written HIP)."""
__version__ = "0.1.0"
