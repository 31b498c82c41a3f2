"""Inter prediction corner cases on the SIMT interpreter against the oracle: pictures of ONE block size (only 8x8 PBs at 10 bits — the densest job lists —, only
64x64 PBs — 128 jobs each, every second one cut by a workgroup boundary once the one-list / two-list / weighted job ranges interleave —, 16x16, 32x32), explicit
weights on half of the blocks (lanes with one and two lists in one workgroup), 8-bit and 12-bit planes, missing references, pictures that end in partial CTBs.
(Written for the LDS-window variant of k_inter_jobs, tools/experiments/inter_windows_through_lds.patch — where a mutant without its rounds fails five of them —
and kept for the per-lane kernel: they are the job-list shapes the mixed synthetic pictures do not contain.)"""
import pytest

from oracle_py import Oracle
from synth_util import assert_planes_equal, device_decode, make_case, oracle_decode
from test_emu_picture import emu_lib  # noqa: F401  (fixture)
from libde265_amd import capi

CASES = [
    dict(width=256, height=192, bit_depth=10, seed=6101, fixed_cu_log2=3, oob_mv_pct=0, intra_pct=0, weighted_pct=0),
    dict(width=256, height=192, bit_depth=10, seed=6102, fixed_cu_log2=3, oob_mv_pct=10, intra_pct=5, weighted_pct=50),
    dict(width=256, height=192, bit_depth=8, seed=6103, fixed_cu_log2=3, oob_mv_pct=0, intra_pct=0, weighted_pct=20),
    dict(width=512, height=256, bit_depth=10, seed=6104, fixed_cu_log2=6, oob_mv_pct=0, intra_pct=0, weighted_pct=30),
    dict(width=512, height=256, bit_depth=8, seed=6105, fixed_cu_log2=6, oob_mv_pct=5, intra_pct=10, weighted_pct=0, bipred_pct=100),
    dict(width=384, height=256, bit_depth=10, seed=6106, fixed_cu_log2=5, oob_mv_pct=0, intra_pct=0, bipred_pct=0),
    dict(width=384, height=256, bit_depth=12, seed=6107, fixed_cu_log2=4, oob_mv_pct=2, intra_pct=5, weighted_pct=50),
    dict(width=328, height=200, bit_depth=10, seed=6108, oob_mv_pct=3, intra_pct=5, weighted_pct=25),
    dict(width=328, height=200, bit_depth=8, seed=6109, oob_mv_pct=3, intra_pct=5, weighted_pct=25, chroma_format=4),
    dict(width=640, height=192, bit_depth=9, seed=6110, fixed_cu_log2=3, oob_mv_pct=0, intra_pct=0, bipred_pct=100, features=256),
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "%dx%d_%dbit_cu%d_seed%d" % (c["width"], c["height"], c["bit_depth"], c.get("fixed_cu_log2", 0), c["seed"]))
def test_windows_through_lds_emulated(emu_lib, oracle, case):  # noqa: F811
    pic, refs = make_case(**case)
    ctx = capi.Context(emu_lib, 0)
    try:
        assert_planes_equal(device_decode(ctx, pic, refs), oracle_decode(Oracle(oracle), pic, refs), repr(case))
    finally:
        ctx.close()
