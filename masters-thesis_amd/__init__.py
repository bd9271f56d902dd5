"""think-and-tell on MI355X: the fMRI->caption training/decoding hot path of
seang123/Masters-Thesis behind its Keras-style surface, running on hand-written
gfx950 HIP kernels (C ABI: include/tnt_hip.h).  Import as ``masters_thesis_amd``."""
import os as _os

# Kernel arguments in device memory (the HIP runtime reads this switch once, when it initialises -- the first HIP call of the
# process, which ``import torch`` is not).  On this image (ROCm 7.2, gfx950) it is the runtime's default already: setting it to 0
# costs ~2 us per eager launch (the 15 launches of tools/step_breakdown.py 449 -> 483 us; the world-size-1 DP rehearsal, which
# replays launch plans, 0.611 -> 0.639 ms) and ~0.15 us per node of a captured step.  Made explicit for runtimes where it is not
# the default; setdefault: an explicit value in the environment wins.
_os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

from . import _lib  # noqa: F401,E402  (does not load the .so until first use)

__all__ = ["_lib"]
