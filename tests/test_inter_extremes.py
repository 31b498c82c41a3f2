"""Motion compensation on ADVERSARIAL reference content: planes made of the extreme sample values (all maximum, one-sample
checkerboards and stripes, random 0 / max), where the separable filters reach the ends of their int16 intermediates
(fallback-motion.cc:512-626: a two-dimensional 8-tap sum of 0 / 255 samples leaves int16 and is truncated by the
reference's int16 stores) and the weighted-prediction sums leave int16 as well.  k_inter_jobs' lean filters do their shifts
with scaled taps + byte permutes and their write-back in saturating packed 16-bit arithmetic (k_inter.hip): this is the
content that would tell them from the reference's int32 arithmetic if the argument in the kernel's comments were wrong.

CPU tier: oracle == the reference's SCALAR functions (oracle/_ref replay) == the kernels under the SIMT interpreter.
GPU tier: the HIP kernels == oracle."""
import numpy as np
import pytest

from oracle_py import Oracle
from synth_util import assert_planes_equal, device_decode, make_case, oracle_decode
from test_emu_picture import emu_lib  # noqa: F401  (fixture)
from libde265_amd import capi, worklist as W

PATTERNS = ["max", "checker", "vstripes", "hstripes", "random01", "checker2"]
CASES = [
    dict(width=128, height=96, bit_depth=8, seed=501, intra_pct=0, bipred_pct=50, weighted_pct=0, cbf_pct=0, oob_mv_pct=5),
    dict(width=128, height=96, bit_depth=10, seed=502, intra_pct=0, bipred_pct=50, weighted_pct=0, cbf_pct=0, oob_mv_pct=5),
    dict(width=136, height=72, bit_depth=12, seed=503, intra_pct=0, bipred_pct=100, weighted_pct=0, cbf_pct=30, log2_ctb=5),
    dict(width=128, height=96, bit_depth=8, seed=504, intra_pct=0, bipred_pct=50, weighted_pct=60, cbf_pct=40, oob_mv_pct=10),
    dict(width=128, height=96, bit_depth=9, seed=505, intra_pct=10, bipred_pct=0, weighted_pct=0, cbf_pct=100, tile_cols=2),
]


def extreme_planes(refs, bd, pattern, seed):
    out = []
    maxv = (1 << bd) - 1
    rng = np.random.default_rng(seed)
    for planes in refs:
        new = []
        for p in planes:
            h, w = p.shape
            yy, xx = np.mgrid[0:h, 0:w]
            if pattern == "max":
                q = np.full((h, w), maxv)
            elif pattern == "checker":
                q = ((xx + yy) & 1) * maxv
            elif pattern == "vstripes":
                q = (xx & 1) * maxv
            elif pattern == "hstripes":
                q = (yy & 1) * maxv
            elif pattern == "checker2":
                q = (((xx >> 1) + (yy >> 1)) & 1) * maxv
            else:
                q = rng.integers(0, 2, size=(h, w)) * maxv
            new.append(q.astype(p.dtype))
        out.append(new)
    return out


def _ids(c):
    return "%dbit_seed%d" % (c["bit_depth"], c["seed"])


@pytest.mark.parametrize("pattern", PATTERNS)
@pytest.mark.parametrize("case", CASES, ids=_ids)
def test_extreme_references_emulated(emu_lib, oracle, ref, case, pattern):  # noqa: F811
    pic, refs = make_case(**case)
    refs = extreme_planes(refs, case["bit_depth"], pattern, case["seed"])
    from ref_replay_py import ref_replay
    st = W.STAGE_INTER | W.STAGE_RESIDUAL | W.STAGE_INTRA
    want = oracle_decode(Oracle(oracle), pic, refs, st)
    assert_planes_equal(want, ref_replay(ref, pic, refs, st, accel=0), "oracle vs scalar reference (%s)" % pattern)
    ctx = capi.Context(emu_lib, 0)
    try:
        for depth in (1, 3):
            ctx.set_pipeline_depth(depth)
            assert_planes_equal(device_decode(ctx, pic, refs, st), want, "kernels vs oracle (%s, depth %d)" % (pattern, depth))
    finally:
        ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("pattern", PATTERNS)
@pytest.mark.parametrize("case", CASES, ids=_ids)
def test_extreme_references_gpu(oracle, case, pattern):
    lib = capi.Library()
    pic, refs = make_case(**case)
    refs = extreme_planes(refs, case["bit_depth"], pattern, case["seed"])
    want = oracle_decode(Oracle(oracle), pic, refs)
    ctx = capi.Context(lib, 0)
    try:
        for depth in (1, 3):
            ctx.set_pipeline_depth(depth)
            assert_planes_equal(device_decode(ctx, pic, refs), want, "HIP vs oracle (%s, depth %d)" % (pattern, depth))
    finally:
        ctx.close()
