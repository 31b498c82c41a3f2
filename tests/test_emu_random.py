"""CPU-tier counterpart of test_gpu_random.py: the same randomised pictures (every third seed; all 200 take 70 s and pass) through the
product kernels under the SIMT interpreter, against the oracle."""
import pytest

from oracle_py import Oracle
from synth_util import assert_planes_equal, device_decode, make_case, oracle_decode
from test_emu_picture import emu_lib  # noqa: F401  (fixture)
from test_gpu_random import random_case
from libde265_amd import capi


@pytest.mark.parametrize("seed", range(0, 200, 3))
def test_random_pictures_emulated(emu_lib, oracle, seed):  # noqa: F811
    pic, refs = make_case(**random_case(seed))
    ctx = capi.Context(emu_lib, 0)
    try:
        assert_planes_equal(device_decode(ctx, pic, refs), oracle_decode(Oracle(oracle), pic, refs), "seed %d" % seed)
    finally:
        ctx.close()
