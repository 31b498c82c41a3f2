"""The SLOT layer under the reference's own decoder: init_acceleration_functions_mi355x() layered on the live decoder's
table at the place the reference layers its SSE / AVX tables (base_context::set_acceleration_functions, decctx.cc:239-270) —
all glue, border construction, dequantiser and filters' decisions stay the reference's CPU code, every table slot it calls
(transforms, qpel / epel, weighted prediction, intra predictors, 8-bit deblocking) runs as a HIP kernel on one block.
One PCIe round trip per block: a compatibility / parity entry, not the fast path (that is tests/test_glue_live.py).

CPU tier: the first pictures of girlshy with the SIMT-interpreter build behind the table, against the reference's own
scalar decode of the same pictures.  GPU tier: the whole stream, against the CI golden MD5 (scripts/ci-run.sh:91-92)."""
import ctypes
import os

import pytest

import de265_py
from test_emu_picture import emu_lib  # noqa: F401  (fixture)
from libde265_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STREAM = os.path.join(ROOT, "glue", "_build", "testdata", "girlshy.h265")
GOLDEN = "b81538fa33a67278e5263e231e43ca98"


def decode_with_table(ref, backend_lib, max_frames=None, threads=0):
    if not os.path.exists(STREAM):
        pytest.skip("glue/_build/testdata not available here")
    ref.ref_layer_acceleration.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    ref.ref_layer_acceleration.restype = ctypes.c_int

    def layer(ctx):
        rc = ref.ref_layer_acceleration(ctx, ctypes.cast(backend_lib.init_acceleration_functions_mi355x, ctypes.c_void_p))
        assert rc == 0, "init_acceleration_functions_mi355x failed (%d): no device?" % rc

    return de265_py.decode_stream(ref, open(STREAM, "rb").read(), after_create=layer, max_frames=max_frames, threads=threads)


def test_reference_decoder_on_emulated_slot_table(ref, emu_lib):  # noqa: F811
    got = decode_with_table(ref, emu_lib.lib, max_frames=4)
    want = de265_py.decode_stream(ref, open(STREAM, "rb").read(), scalar=True, max_frames=4)
    assert got == want and got[1] == 4


@pytest.mark.gpu
def test_reference_decoder_on_gpu_slot_table(ref):
    lib = capi.Library()
    assert lib.device_count() >= 1
    md5, n, warnings = decode_with_table(ref, lib.lib, threads=4)      # the slots are re-entrant (acceleration.h: 32 pool threads)
    assert n == 75 and not warnings and md5 == GOLDEN
