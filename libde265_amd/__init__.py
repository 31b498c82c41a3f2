"""libde265_amd — MI355X-native HEVC pixel-reconstruction backend (host layer).

The compute path is the hand-written HIP library libde265_mi355x.so (csrc/); this package only
binds its C ABI (capi), mirrors the work-list PODs (worklist) and generates synthetic work lists for
benchmarks (synth).  Importing the package never touches the GPU; creating a `capi.Context` does and
raises if the library or the device is missing — there is no CPU fallback."""
from . import worklist  # noqa: F401

__all__ = ["worklist", "capi", "synth"]
