"""Tile sharding on the CPU tier: the product kernels under the SIMT interpreter, N virtual
ranks in one process, against the oracle's whole-picture decode.  Covers deblocking and SAO across
tile boundaries (halo exchange), boundaries with filtering across tiles disabled, more tiles than
ranks, ranks without tiles, 10-bit and all-intra pictures."""
import pytest

from oracle_py import Oracle
from shard_util import group_sharded_decode, local_sharded_decode
from synth_util import assert_planes_equal, make_case, oracle_decode
from test_emu_picture import emu_lib  # noqa: F401  (fixture)
from libde265_amd import worklist

CASES = [
    (dict(width=256, height=128, bit_depth=8, seed=41, tile_cols=2, tile_rows=1), 2),
    (dict(width=256, height=192, bit_depth=8, seed=42, tile_cols=2, tile_rows=2), 4),
    (dict(width=256, height=192, bit_depth=10, seed=43, tile_cols=2, tile_rows=2), 2),
    (dict(width=320, height=128, bit_depth=8, seed=44, tile_cols=3, tile_rows=1, lf_across_tiles=0), 3),
    (dict(width=192, height=192, bit_depth=8, seed=45, tile_cols=1, tile_rows=3, intra_pct=100, n_refs=0), 2),
    (dict(width=200, height=136, bit_depth=8, seed=46, tile_cols=2, tile_rows=2, log2_ctb=4), 3),
    (dict(width=128, height=128, bit_depth=8, seed=47, tile_cols=2, tile_rows=1), 4),          # ranks 1 and 3 own nothing
]


@pytest.mark.parametrize("case,nranks", CASES, ids=lambda v: ("%dx%d_seed%d" % (v["width"], v["height"], v["seed"])) if isinstance(v, dict) else "r%d" % v)
def test_sharded_matches_oracle(emu_lib, oracle, case, nranks):  # noqa: F811
    o = Oracle(oracle)
    pic, refs = make_case(**case)
    want = oracle_decode(o, pic, refs)
    for r, got in enumerate(local_sharded_decode(emu_lib, pic, refs, nranks)):
        assert_planes_equal(got, want, "rank %d of %d" % (r, nranks))


def test_sharded_stage_isolation(emu_lib, oracle):  # noqa: F811
    o = Oracle(oracle)
    pic, refs = make_case(width=256, height=128, bit_depth=8, seed=48, tile_cols=2, tile_rows=2)
    for st in (worklist.STAGE_INTER | worklist.STAGE_RESIDUAL | worklist.STAGE_INTRA,
               worklist.STAGE_ALL & ~worklist.STAGE_SAO):
        want = oracle_decode(o, pic, refs, st)
        for r, got in enumerate(local_sharded_decode(emu_lib, pic, refs, 2, stages=st)):
            assert_planes_equal(got, want, "stages %d rank %d" % (st, r))


GROUP_CASES = [(CASES[0][0], 2), (CASES[1][0], 4), (CASES[3][0], 3), (CASES[6][0], 4), (CASES[2][0], 1)]


@pytest.mark.parametrize("case,nranks", GROUP_CASES, ids=lambda v: ("%dx%d_seed%d" % (v["width"], v["height"], v["seed"])) if isinstance(v, dict) else "r%d" % v)
def test_group_in_one_process_matches_oracle(emu_lib, oracle, case, nranks):  # noqa: F811
    """m355_group_*: the ranks are contexts of ONE process, the exchanges copies between their buffers (no callbacks, no
    collective library) — what a decoder with a parser thread per tile uses to spread one bitstream over the GPUs of a node"""
    o = Oracle(oracle)
    pic, refs = make_case(**case)
    want = oracle_decode(o, pic, refs)
    for r, got in enumerate(group_sharded_decode(emu_lib, pic, refs, nranks, repeat=2)):
        assert_planes_equal(got, want, "group rank %d of %d" % (r, nranks))


IN_PLACE_CASES = [(CASES[1][0], 4, 1), (CASES[2][0], 2, 2), (CASES[6][0], 4, 1)]


@pytest.mark.parametrize("case,nranks,depth", IN_PLACE_CASES, ids=lambda v: ("%dx%d_seed%d" % (v["width"], v["height"], v["seed"])) if isinstance(v, dict) else "n%d" % v)
def test_group_lists_recorded_in_place(emu_lib, oracle, case, nranks, depth):  # noqa: F811
    """m355_picture_arena_begin: the in-place path of a tile-sharded context — every rank's lists are written into its handle's
    pinned arena (room for the other ranks' border units behind cus[] / pbs[] sized from the picture parameters) and taken over by
    m355_picture_replace without a host copy; the handles are recorded again before the second round"""
    o = Oracle(oracle)
    pic, refs = make_case(**case)
    want = oracle_decode(o, pic, refs)
    for r, got in enumerate(group_sharded_decode(emu_lib, pic, refs, nranks, depth=depth, repeat=2, in_place=True)):
        assert_planes_equal(got, want, "in-place group rank %d of %d" % (r, nranks))


def test_group_decode_rejects_a_bad_handle_without_hanging(emu_lib):  # noqa: F811
    """one invalid handle used to leave its rank's thread before it published a step, and the neighbours' threads spun
    forever.  Every handle is validated before any rank starts: the call returns an error, and the group still decodes afterwards."""
    from shard_util import _setup_rank
    from libde265_amd import capi
    pic, refs = make_case(width=256, height=192, bit_depth=8, seed=51, tile_cols=2, tile_rows=2)
    ctxs = [capi.Context(emu_lib, 0) for _ in range(2)]
    grp = capi.Group(emu_lib, ctxs)
    try:
        hs = []
        for r, ctx in enumerate(ctxs):
            sp, _dst = _setup_rank(ctx, pic, refs, r, 2, "cpu")
            hs.append(ctx.upload(sp))
        with pytest.raises(capi.M355Error):
            grp.decode([hs[0], 12345], True)
        with pytest.raises(capi.M355Error):
            grp.decode([hs[0], -1], True)
        grp.decode(hs, True)          # still alive
        grp.wait()
    finally:
        grp.close()
        for ctx in ctxs:
            ctx.close()
