"""Second pin of the CPU oracle: whole synthetic pictures replayed through the REAL reference functions
(oracle/ref_replay.cc -> generate_inter_prediction_samples, scale_coefficients, decode_intra_prediction,
apply_deblocking_filter, apply_sample_adaptive_offset_sequential of /root/reference, scalar AND SSE/AVX tables)
must equal oracle/hevc_oracle.c bit for bit — tiles with and without filtering across them, 8/10/12-bit, CTB
16/32/64, every CU/TU size, explicit weights, MVs far outside the picture, transform skip, intra-only pictures,
several slices with their own filter flags, constrained intra prediction, cu_transquant_bypass, scaling lists.
(The first pin is the recorded girlshy stream, tests/test_girlshy_oracle.py.)  Needs oracle/_ref (built from
/root/reference by oracle/Makefile); skipped where that is unavailable."""
import pytest

from oracle_py import Oracle
from ref_replay_py import ref_replay
from synth_util import assert_planes_equal, make_case, oracle_decode
from libde265_amd import synth, worklist as W

CASES = [
    dict(width=192, height=128, bit_depth=8, seed=11),
    dict(width=200, height=136, bit_depth=8, seed=12, tile_cols=2, tile_rows=2, lf_across_tiles=0),
    dict(width=192, height=128, bit_depth=10, seed=13, tile_cols=3, tile_rows=1),
    dict(width=128, height=128, bit_depth=8, seed=14, intra_pct=100, n_refs=0, tile_cols=2, tile_rows=1),
    dict(width=136, height=72, bit_depth=12, seed=15, log2_ctb=5, intra_pct=40),
    dict(width=128, height=64, bit_depth=8, seed=16, log2_ctb=4, fixed_cu_log2=3, cbf_pct=100),
    dict(width=416, height=240, bit_depth=8, seed=21),
    dict(width=832, height=480, bit_depth=10, seed=22, tile_cols=3, tile_rows=2),
    dict(width=640, height=368, bit_depth=10, seed=24, intra_pct=100, n_refs=0, tile_cols=2, tile_rows=2),
    dict(width=1280, height=720, bit_depth=8, seed=25, intra_pct=25, weighted_pct=50, oob_mv_pct=20),
    dict(width=416, height=240, bit_depth=12, seed=28, weighted_pct=30, oob_mv_pct=10),
    dict(width=64, height=64, bit_depth=8, seed=30, oob_mv_pct=100, intra_pct=0),
    dict(width=72, height=24, bit_depth=9, seed=27, log2_ctb=4, intra_pct=50),
    # several slices (random deblock-disable / filter-across-slices / SAO flags, beta / tc offsets)
    dict(width=256, height=192, bit_depth=8, seed=71, n_slices=4),
    dict(width=256, height=192, bit_depth=10, seed=72, n_slices=3, tile_cols=2, tile_rows=2),
    dict(width=384, height=256, bit_depth=8, seed=77, n_slices=6, tile_cols=3, tile_rows=2, lf_across_tiles=0, intra_pct=30),
    # constrained_intra_pred, cu_transquant_bypass CUs, scaling lists; and all of it together
    dict(width=256, height=192, bit_depth=8, seed=73, features=1, intra_pct=40),
    dict(width=256, height=192, bit_depth=8, seed=74, features=2),
    dict(width=256, height=192, bit_depth=10, seed=75, features=4),
    dict(width=320, height=192, bit_depth=8, seed=76, features=7, n_slices=5, intra_pct=30),
    dict(width=320, height=192, bit_depth=12, seed=78, features=7, n_slices=3, intra_pct=100, n_refs=0, log2_ctb=5),
    # PCM coding units, with and without pcm_loop_filter_disable
    dict(width=256, height=192, bit_depth=8, seed=81, features=8, intra_pct=50),
    dict(width=256, height=192, bit_depth=10, seed=82, features=8 + 16 + 2, intra_pct=50, n_slices=3),
    dict(width=256, height=192, bit_depth=8, seed=83, features=31, intra_pct=100, n_refs=0),
    # monochrome, 4:4:4, 4:2:2 (range extensions), alone and with everything above
    dict(width=256, height=192, bit_depth=8, seed=91, chroma_format=4),
    dict(width=256, height=192, bit_depth=8, seed=92, chroma_format=3, intra_pct=30),
    dict(width=256, height=192, bit_depth=10, seed=93, chroma_format=2, intra_pct=30),
    dict(width=256, height=192, bit_depth=8, seed=94, chroma_format=3, features=31, n_slices=3, tile_cols=2, intra_pct=50),
    dict(width=256, height=192, bit_depth=10, seed=95, chroma_format=2, features=31, n_slices=3, tile_rows=2, intra_pct=50),
    # cross-component prediction (4:4:4), incl. chroma blocks without coefficients and bypass / scaling-list CUs
    dict(width=256, height=192, bit_depth=8, seed=101, chroma_format=3, features=32, intra_pct=30),
    dict(width=256, height=192, bit_depth=10, seed=102, chroma_format=3, features=32 + 31, intra_pct=50, n_slices=2),
    # range-extension residual tools: transform skip up to 32x32 with implicit (intra 10 / 26) and explicit (inter) RDPCM on skip
    # and bypass blocks (64), transform_skip_rotation of 4x4 blocks (128); both, with bypass CUs, in 4:2:0, 4:4:4 and 4:2:2
    dict(width=256, height=192, bit_depth=8, seed=111, features=64, intra_pct=40, cbf_pct=90),
    dict(width=256, height=192, bit_depth=10, seed=112, features=64 + 2, intra_pct=50, cbf_pct=90, fixed_cu_log2=3),
    dict(width=256, height=192, bit_depth=8, seed=113, features=128 + 2, intra_pct=60, cbf_pct=100, fixed_cu_log2=3),
    dict(width=256, height=192, bit_depth=12, seed=114, features=64 + 128 + 2, chroma_format=3, intra_pct=50, cbf_pct=90),
    dict(width=256, height=192, bit_depth=8, seed=115, features=64 + 128 + 2 + 4, chroma_format=2, intra_pct=50, cbf_pct=90, n_slices=2),
    # prediction blocks whose reference picture is missing (motion.cc:362-376): uni, bi (one or both lists), weighted
    dict(width=256, height=192, bit_depth=8, seed=121, features=256, intra_pct=5, weighted_pct=30),
    dict(width=256, height=192, bit_depth=10, seed=122, features=256, intra_pct=0, bipred_pct=100, weighted_pct=0, tile_cols=2),
    dict(width=256, height=192, bit_depth=8, seed=123, features=256, intra_pct=0, chroma_format=3, bipred_pct=50),
    # blocks that carry already-scaled levels (M355_RBF_DEQUANTIZED): the reference entered below its dequantiser
    dict(width=256, height=192, bit_depth=8, seed=131, features=512, intra_pct=30, cbf_pct=90),
    dict(width=256, height=192, bit_depth=10, seed=132, features=512 + 64 + 128, intra_pct=50, cbf_pct=90),
]
STAGES = [W.STAGE_ALL, W.STAGE_INTER | W.STAGE_RESIDUAL | W.STAGE_INTRA, W.STAGE_ALL & ~W.STAGE_SAO, W.STAGE_INTER]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "%dx%d_%dbit_seed%d" % (c["width"], c["height"], c["bit_depth"], c["seed"]))
def test_oracle_equals_reference_replay(oracle, ref, case):
    o = Oracle(oracle)
    pic, refs = make_case(**case)
    for st in STAGES:
        want = ref_replay(ref, pic, refs, st, accel=0)
        assert_planes_equal(oracle_decode(o, pic, refs, st), want, "oracle vs scalar reference, stages %d" % st)
    # the reference's SIMD tables give the same picture as its scalar ones (and hence as the oracle)
    assert_planes_equal(ref_replay(ref, pic, refs, W.STAGE_ALL, accel=1), ref_replay(ref, pic, refs, W.STAGE_ALL, accel=0), "SSE vs scalar reference")


@pytest.mark.parametrize("name", ["c2_1080p_intra", "c3_4k_inter", "c4_4k_4tiles", "c5_8k10_8tiles"])   # every BASELINE config at full size, the headline list included
def test_oracle_equals_reference_replay_baseline_configs(oracle, ref, name):
    o = Oracle(oracle)
    pic, refs = make_case(**synth.CONFIGS[name])
    assert_planes_equal(oracle_decode(o, pic, refs), ref_replay(ref, pic, refs, accel=1), name)
