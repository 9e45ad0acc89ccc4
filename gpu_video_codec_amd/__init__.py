"""MI355X-native HEVC in-loop deblocking filter (drop-in for the hot path of gpu_video_codec).

Product = gpu_video_codec_amd/libhevcdbk.so (C ABI in include/hevc_deblock.h; hand-written HIP for
gfx950 under csrc/).  The Python modules are bindings/plumbing for tests and bench.py.
"""
from . import _lib  # noqa: F401  (does not load the library until first use)

__all__ = ["deblock", "synth"]
