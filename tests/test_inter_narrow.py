"""Pictures narrower than the vector loads of k_inter_jobs' lean EDGE path (12 / 16 luma, 6 / 8 chroma samples of a row in one load): m355_launch_inter keeps
the general kernels for them (k_inter.hip `wide`).  8- and 16-sample-wide pictures, where every window leaves the picture on both sides at once, in 8 / 10 /
12 bit, with explicit weights and two lists — and the first widths that DO take the lean kernels, with most vectors pointing outside (the LDS row extension at
both picture borders)."""
import pytest

from oracle_py import Oracle
from synth_util import assert_planes_equal, device_decode, make_case, oracle_decode
from test_emu_picture import emu_lib  # noqa: F401  (fixture)
from libde265_amd import capi

CASES = [
    dict(width=8, height=8, bit_depth=8, seed=3, log2_ctb=4, oob_mv_pct=50, intra_pct=0),
    dict(width=8, height=24, bit_depth=10, seed=4, log2_ctb=4, oob_mv_pct=50, intra_pct=0, bipred_pct=100),
    dict(width=8, height=16, bit_depth=12, seed=7, log2_ctb=4, oob_mv_pct=100, intra_pct=0, weighted_pct=60),
    dict(width=16, height=8, bit_depth=8, seed=5, log2_ctb=4, oob_mv_pct=80, intra_pct=0),
    dict(width=16, height=16, bit_depth=10, seed=8, log2_ctb=4, oob_mv_pct=100, intra_pct=0, bipred_pct=100, weighted_pct=50),
    dict(width=24, height=16, bit_depth=12, seed=6, log2_ctb=4, oob_mv_pct=80, intra_pct=0, bipred_pct=100, weighted_pct=50),
    dict(width=32, height=24, bit_depth=8, seed=9, log2_ctb=5, oob_mv_pct=100, intra_pct=0, bipred_pct=50),
]


def _ids(c):
    return "%dx%d_%dbit" % (c["width"], c["height"], c["bit_depth"])


@pytest.mark.parametrize("case", CASES, ids=_ids)
def test_narrow_pictures_emulated(emu_lib, oracle, case):  # noqa: F811
    pic, refs = make_case(**case)
    ctx = capi.Context(emu_lib, 0)
    try:
        assert_planes_equal(device_decode(ctx, pic, refs), oracle_decode(Oracle(oracle), pic, refs), "kernels vs oracle %r" % (case,))
    finally:
        ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=_ids)
def test_narrow_pictures_gpu(oracle, case):
    pic, refs = make_case(**case)
    ctx = capi.Context(capi.Library(), 0)
    try:
        want = oracle_decode(Oracle(oracle), pic, refs)
        for depth in (1, 3):
            ctx.set_pipeline_depth(depth)
            assert_planes_equal(device_decode(ctx, pic, refs), want, "HIP vs oracle %r depth %d" % (case, depth))
    finally:
        ctx.close()
