"""MI355X-native visual-odometry front end (detect -> describe -> match -> pose).

The compute path is libvo_hip.so (hand-written HIP kernels for gfx950 behind
the C ABI of include/vo_hip.h); `capi` binds it, `synth` generates seeded
synthetic sequences.  Importing the package does not touch the GPU.
"""
from . import capi, synth  # noqa: F401

__all__ = ["capi", "synth"]
