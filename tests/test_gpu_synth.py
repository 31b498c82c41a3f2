"""GPU parity on synthetic work lists (configs C2..C5 of BASELINE.json): HIP kernels through the C ABI
vs the oracle, bit-exact, from small edge-case pictures up to BASELINE's full sizes (4K 8-bit, 8K
10-bit 8 tiles), plus size-independent properties at full size (determinism under replay, stage
composition = the reference's DISABLE_DEBLOCKING / DISABLE_SAO isolation)."""
import numpy as np
import pytest

from oracle_py import Oracle
from synth_util import assert_planes_equal, device_decode, make_case, oracle_decode
from test_emu_synth import CASES
from libde265_amd import capi, synth, worklist

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    lib = capi.Library()
    assert lib.device_count() >= 1
    c = capi.Context(lib, 0)
    yield c
    c.close()


SMALL = CASES + [
    dict(width=416, height=240, bit_depth=8, seed=21),
    dict(width=832, height=480, bit_depth=10, seed=22, tile_cols=3, tile_rows=2),
    dict(width=640, height=360, bit_depth=8, seed=23, intra_pct=100, n_refs=0),
    dict(width=640, height=368, bit_depth=10, seed=24, intra_pct=100, n_refs=0, tile_cols=2, tile_rows=2),
    dict(width=1280, height=720, bit_depth=8, seed=25, intra_pct=25, weighted_pct=50, oob_mv_pct=20),
    dict(width=8, height=8, bit_depth=8, seed=26, log2_ctb=4),
    dict(width=72, height=24, bit_depth=9, seed=27, log2_ctb=4, intra_pct=50),
    dict(width=416, height=240, bit_depth=12, seed=28, weighted_pct=30, oob_mv_pct=10),
    dict(width=416, height=240, bit_depth=16, seed=29, weighted_pct=30, oob_mv_pct=10, intra_pct=0, cbf_pct=0),
    dict(width=64, height=64, bit_depth=8, seed=30, oob_mv_pct=100, intra_pct=0),
]


@pytest.mark.parametrize("case", SMALL, ids=lambda c: "%dx%d_%dbit_seed%d" % (c["width"], c["height"], c["bit_depth"], c["seed"]))
def test_synthetic_small(ctx, oracle, case):
    o = Oracle(oracle)
    pic, refs = make_case(**case)
    assert_planes_equal(device_decode(ctx, pic, refs), oracle_decode(o, pic, refs), "all stages")
    st = worklist.STAGE_INTER | worklist.STAGE_RESIDUAL | worklist.STAGE_INTRA
    assert_planes_equal(device_decode(ctx, pic, refs, st), oracle_decode(o, pic, refs, st), "no loop filters")
    st |= worklist.STAGE_DEBLOCK
    assert_planes_equal(device_decode(ctx, pic, refs, st), oracle_decode(o, pic, refs, st), "no SAO")


@pytest.mark.parametrize("name", ["c2_1080p_intra", "c3_4k_inter", "c4_4k_4tiles", "c5_8k10_8tiles"])
def test_baseline_configs_full_size(ctx, oracle, name):
    o = Oracle(oracle)
    pic, refs = make_case(**synth.CONFIGS[name])
    want = oracle_decode(o, pic, refs)
    got = device_decode(ctx, pic, refs, resident=True)
    assert_planes_equal(got, want, name)
    # determinism: replaying the resident work lists many times gives the same picture
    again = device_decode(ctx, pic, refs, resident=True, repeat=5)
    assert_planes_equal(again, want, name + " (replayed x5)")


def test_empty_picture(ctx, oracle):
    """no coded blocks at all: output must stay at the frame's initial zero (image.cc:164)"""
    pic, refs = make_case(width=64, height=64, bit_depth=8, seed=5, n_refs=0, intra_pct=100)
    for name in ("cus", "tus", "pbs", "wts", "rbs", "ibs", "coeffs"):
        setattr(pic, name, getattr(pic, name)[:0])
    pic.rb_count = [0, 0, 0, 0]; pic.res_len = 0
    pic.ctbs["ib_start"] = 0; pic.ctbs["ib_count"] = 0; pic.ctbs["sao_type"] = 0
    got = device_decode(ctx, pic, [])
    assert all(int(p.max()) == 0 for p in got)


def test_malformed_lists_are_rejected(ctx):
    """the boundary must be total: bad indices are refused, never sent to the GPU"""
    pic, refs = make_case(width=128, height=64, bit_depth=8, seed=6)
    bad = pic.pbs.copy(); bad["x"][0] = 5000
    pic2 = pic; saved = pic.pbs; pic2.pbs = bad
    with pytest.raises(capi.M355Error):
        device_decode(ctx, pic2, refs)
    pic.pbs = saved
