"""MI355X-native inter-frame path of the vp8oclenc VP8 encoder (hand-written HIP behind a C ABI).

Layout: csrc/ (HIP kernels, C ABI, host mirror), api.py (ctypes binding), driver.py (the reference's
frame loop reduced to this path), synth.py (seeded synthetic YUV), build.py (hipcc, in-tree).
"""
from .api import Vp8Hip, Vp8HipError, load_library  # noqa: F401
