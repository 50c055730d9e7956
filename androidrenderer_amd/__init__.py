"""androidrenderer_amd — MI355X-native deferred lighting + GI + post hot path of the SAH renderer.

The product is libsah_hip.so (hand-written HIP kernels for gfx950 behind the C ABI in include/sah_hip.h).
This package holds its sources (csrc/), the build recipe, a ctypes binding and host-side producers of the
uniform blocks; importing it does not load the library — `lib.load()` / `lib.Context()` do, and fail loudly
if it is missing."""
from . import _abi, images, scene, synth  # noqa: F401
from . import lib  # noqa: F401

__all__ = ["_abi", "images", "scene", "synth", "lib"]
