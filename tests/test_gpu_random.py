"""GPU parity, randomised: picture geometry, bit depth, chroma format, CTB size, tiling, slicing, block mix and the optional
coding tools are drawn at random (seeded), the work lists come from the synthetic generator, and the HIP kernels must
reproduce the oracle bit for bit — with one picture at a time and with three in flight.  The same generator output was
pinned against the real reference's functions on the CPU tier (test_oracle_vs_ref_replay.py); this sweep is about the
hardware: scheduling, memory ordering and occupancy effects that the SIMT interpreter cannot show."""
import numpy as np
import pytest

from oracle_py import Oracle
from synth_util import assert_planes_equal, device_decode, make_case, oracle_decode
from libde265_amd import capi, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    lib = capi.Library()
    assert lib.device_count() >= 1
    c = capi.Context(lib, 0)
    yield c
    c.close()


def random_case(seed):
    rng = np.random.default_rng(7000 + seed)
    log2_ctb = int(rng.choice([4, 5, 6, 6]))
    mincb = 8
    big = seed >= 48                                     # the first 48 stay small (the CPU tier replays a subset under the interpreter)
    w = int(rng.integers(2, 81 if big else 40)) * mincb
    h = int(rng.integers(2, 46 if big else 24)) * mincb
    cf = int(rng.choice([0, 0, 0, 2, 3, 4]))            # 0/1 = 4:2:0, 2 = 4:2:2, 3 = 4:4:4, 4 = monochrome
    bd = int(rng.choice([8, 8, 9, 10, 10, 12]))
    ctbs_x, ctbs_y = -(-w >> log2_ctb), -(-h >> log2_ctb)
    tc = int(rng.integers(1, min(4, ctbs_x) + 1)) if rng.random() < 0.5 else 1
    tr = int(rng.integers(1, min(3, ctbs_y) + 1)) if rng.random() < 0.5 else 1
    feats = 0
    for bit, pr in ((synth.SYN_CONSTRAINED_INTRA, .3), (synth.SYN_TRANSQUANT_BYPASS, .3), (synth.SYN_SCALING_LIST, .3), (synth.SYN_PCM, .3),
                    (synth.SYN_PCM_LOOP_FILTER_DISABLE, .2), (synth.SYN_RDPCM, .25), (synth.SYN_ROTATE, .25), (synth.SYN_MISSING_REF, .25),
                    (synth.SYN_DEQUANTIZED, .15)):
        if rng.random() < pr:
            feats |= bit
    if cf == 3 and rng.random() < 0.5:
        feats |= synth.SYN_CROSS_COMPONENT
        feats &= ~synth.SYN_DEQUANTIZED                    # pre-scaled levels come from the table slots, which cross-component pictures bypass
    intra = int(rng.choice([0, 5, 30, 100]))
    return dict(width=w, height=h, bit_depth=bd, log2_ctb=log2_ctb, tile_cols=tc, tile_rows=tr, intra_pct=intra,
                n_refs=0 if intra == 100 else 2, bipred_pct=int(rng.choice([0, 50, 100])), weighted_pct=int(rng.choice([0, 10, 60])),
                oob_mv_pct=int(rng.choice([0, 2, 30])), cbf_pct=int(rng.choice([0, 40, 100])), deblock=int(rng.random() < 0.8),
                sao=int(rng.random() < 0.8), lf_across_tiles=int(rng.random() < 0.6), n_slices=int(rng.choice([0, 0, 2, 5])),
                features=feats, chroma_format=cf, seed=9000 + seed)


@pytest.mark.parametrize("seed", range(200))
def test_random_pictures_bit_exact(ctx, oracle, seed):
    case = random_case(seed)
    try:
        pic, refs = make_case(**case)
    except RuntimeError as e:              # a combination the generator does not build (it says so) is not a test failure
        pytest.skip("generator: %s" % e)
    want = oracle_decode(Oracle(oracle), pic, refs)
    ctx.set_pipeline_depth(1)
    assert_planes_equal(device_decode(ctx, pic, refs), want, "depth 1: %r" % (case,))
    ctx.set_pipeline_depth(3)
    try:
        assert_planes_equal(device_decode(ctx, pic, refs, resident=True, repeat=4), want, "depth 3: %r" % (case,))
    finally:
        ctx.set_pipeline_depth(1)
