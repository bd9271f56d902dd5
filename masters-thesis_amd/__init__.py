"""think-and-tell on MI355X: the fMRI->caption training/decoding hot path of
seang123/Masters-Thesis behind its Keras-style surface, running on hand-written
gfx950 HIP kernels (C ABI: include/tnt_hip.h).  Import as ``masters_thesis_amd``."""
from . import _lib  # noqa: F401  (does not load the .so until first use)

__all__ = ["_lib"]
