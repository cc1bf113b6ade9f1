"""cvo_slam_amd -- MI355X (gfx950) implementation of CVO-SLAM's per-frame-pair CVO
alignment hot path (thirdparty/cvo of bexilin/CVO-SLAM), behind the C ABI of
include/cvo_hip.h.

Layout
  csrc/       hand-written HIP kernels + the C-ABI implementation (libcvo_hip.so)
  api.py      ctypes mirror of the reference's `cvo::cvo` call surface (Cvo, CvoBatch)
  synth.py    seeded synthetic RGB-D pairs (bench + tests)
  build.py    hipcc driver (gfx950 only)

The product path never imports oracle/ and has no CPU fallback: if libcvo_hip.so is
missing or no gfx950 device is visible, calls raise.
"""
from .api import Cvo, CvoBatch, CvoComm, CvoMulti, CvoError, default_params, device_count, lib_path, load_library  # noqa: F401
