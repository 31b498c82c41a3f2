"""Kernel-logic verification (CPU tier) on synthetic pictures: tiles, 10-bit, every CU size, explicit
weights, out-of-picture MVs, transform skip — through the SIMT-interpreter build vs the oracle."""
import pytest

from oracle_py import Oracle
from synth_util import assert_planes_equal, device_decode, make_case, oracle_decode
from test_emu_picture import emu_lib  # noqa: F401  (fixture)
from libde265_amd import capi, worklist

CASES = [
    dict(width=192, height=128, bit_depth=8, seed=11),
    dict(width=200, height=136, bit_depth=8, seed=12, tile_cols=2, tile_rows=2, lf_across_tiles=0),
    dict(width=192, height=128, bit_depth=10, seed=13, tile_cols=3, tile_rows=1),
    dict(width=128, height=128, bit_depth=8, seed=14, intra_pct=100, n_refs=0, tile_cols=2, tile_rows=1),
    dict(width=136, height=72, bit_depth=12, seed=15, log2_ctb=5, intra_pct=40),
    dict(width=128, height=64, bit_depth=8, seed=16, log2_ctb=4, fixed_cu_log2=3, cbf_pct=100),
    dict(width=192, height=128, bit_depth=8, seed=17, n_slices=4, features=7, intra_pct=30),
    dict(width=192, height=128, bit_depth=10, seed=18, n_slices=3, tile_cols=2, tile_rows=2, features=7),
    dict(width=192, height=128, bit_depth=8, seed=19, features=31, intra_pct=60, n_slices=2),
    dict(width=128, height=128, bit_depth=8, seed=31, chroma_format=4),
    dict(width=128, height=128, bit_depth=8, seed=32, chroma_format=3, intra_pct=30, features=31),
    dict(width=128, height=128, bit_depth=10, seed=33, chroma_format=2, intra_pct=30, features=31, tile_cols=2),
    dict(width=128, height=128, bit_depth=8, seed=34, chroma_format=3, intra_pct=30, features=32),
    dict(width=192, height=128, bit_depth=10, seed=35, chroma_format=3, intra_pct=40, features=63, log2_ctb=5),
    # RDPCM (64) / rotation (128) on skip + bypass blocks, missing references (256), pre-scaled levels (512)
    dict(width=192, height=128, bit_depth=8, seed=41, features=64 + 128 + 2, intra_pct=50, cbf_pct=90),
    dict(width=128, height=128, bit_depth=10, seed=42, features=64 + 128 + 2, chroma_format=3, intra_pct=50, cbf_pct=90, fixed_cu_log2=3),
    dict(width=192, height=128, bit_depth=8, seed=43, features=256, intra_pct=5, weighted_pct=30),
    dict(width=128, height=128, bit_depth=10, seed=44, features=512 + 64 + 128, intra_pct=40, cbf_pct=90, chroma_format=2),
    # intra pictures (k_intra's 12-wave kernel: persistent workgroups, exec records, register-resident 4x4 / 8x8 path, batched
    # 16x16 / 32x32 loops) in every chroma format, with raw / bypass / constrained-intra blocks, one CU size at a time and mixed
    dict(width=128, height=128, bit_depth=8, seed=51, intra_pct=100, n_refs=0, chroma_format=3, features=31),
    dict(width=192, height=128, bit_depth=12, seed=52, intra_pct=100, n_refs=0, chroma_format=2, features=31, tile_cols=2),
    dict(width=128, height=128, bit_depth=10, seed=53, intra_pct=100, n_refs=0, chroma_format=4),
    dict(width=128, height=128, bit_depth=8, seed=54, intra_pct=100, n_refs=0, fixed_cu_log2=6),
    dict(width=128, height=128, bit_depth=10, seed=55, intra_pct=100, n_refs=0, fixed_cu_log2=5, chroma_format=3),
    dict(width=136, height=72, bit_depth=8, seed=56, intra_pct=100, n_refs=0, fixed_cu_log2=4, n_slices=2),
    dict(width=128, height=128, bit_depth=8, seed=57, intra_pct=100, n_refs=0, fixed_cu_log2=3, cbf_pct=0),
    # inter pictures whose intra blocks are 32x32 (the chains of such a picture): a CTB's luma waves share the block, its residual comes
    # from LDS as in an intra picture's kernel — 4:2:0, 4:4:4 (32x32 chroma blocks, shared too), 4:2:2, with and without residuals
    dict(width=256, height=128, bit_depth=8, seed=61, intra_pct=50, fixed_cu_log2=5, cbf_pct=100),
    dict(width=192, height=128, bit_depth=10, seed=62, intra_pct=40, fixed_cu_log2=6, chroma_format=3, cbf_pct=80),
    dict(width=192, height=128, bit_depth=8, seed=63, intra_pct=60, fixed_cu_log2=5, chroma_format=2, tile_cols=2),
    dict(width=256, height=192, bit_depth=10, seed=64, intra_pct=30, fixed_cu_log2=5, cbf_pct=0, features=31),
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "%dx%d_%dbit_seed%d" % (c["width"], c["height"], c["bit_depth"], c["seed"]))
def test_emulated_kernels_match_oracle(emu_lib, oracle, case):  # noqa: F811
    o = Oracle(oracle)
    pic, refs = make_case(**case)
    want = oracle_decode(o, pic, refs)
    ctx = capi.Context(emu_lib, 0)
    try:
        got = device_decode(ctx, pic, refs)
        assert_planes_equal(got, want, "all stages")
        # stage-isolated: prediction + residual only (the DISABLE_DEBLOCKING / DISABLE_SAO oracle of de265.h:409-410)
        st = worklist.STAGE_INTER | worklist.STAGE_RESIDUAL | worklist.STAGE_INTRA
        assert_planes_equal(device_decode(ctx, pic, refs, st), oracle_decode(o, pic, refs, st), "no loop filters")
    finally:
        ctx.close()
