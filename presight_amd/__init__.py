"""presight_amd — MI355X-native (gfx950) kernels + host glue for the PreSight NeRF prior-builder hot path.

The arithmetic lives in libpresight_hip.so (hand-written HIP, C ABI in include/presight_hip.h);
this package mirrors the reference's nerfstudio operator/plugin interface on top of it."""
from ._lib import LIB_PATH, PresightHipError, lib  # noqa: F401

__all__ = ["lib", "LIB_PATH", "PresightHipError"]
