"""MI355X-native A-TVSNet inference hot path.

Host side: Python mirrors of the reference's operator API
(``cnn_wrapper.network`` / ``cnn_wrapper.atvsnet``), model assembly
(``atvsnet.model``), geometry (``atvsnet.homography_warping``) and entry point
(``atvsnet.example``).  Device side: hand-written HIP kernels for gfx950 behind
the C-ABI declared in ``include/atvsnet_hip.h`` (``csrc/``), reached through
``ops`` (ctypes; torch is used for device memory and streams only).
"""
from .flags import FLAGS, AUTO_REUSE          # noqa: F401
from . import variables                       # noqa: F401

__all__ = ['FLAGS', 'AUTO_REUSE', 'variables']
