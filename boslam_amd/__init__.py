"""boslam_amd — MI355X (gfx950) implementation of boslam's front-end + bundle-adjustment hot path.

Everything numeric runs in boslam_amd/csrc/libboslam_hip.so (hand-written HIP kernels behind the C ABI of
include/boslam_hip.h).  The Python modules mirror the third-party call sites of the reference
(cv2 / g2o at camera.py, slam/tracking.py, slam/optimizer.py) so that its orchestration code is untouched."""
from ._lib import Context, BhError, default_context, lib, LIB_PATH  # noqa: F401

__version__ = "0.1.0"
