"""Lists recorded IN PLACE (m355_arena_begin -> the caller writes into the pinned arena -> m355_submit_picture copies nothing on
the host): same picture as the copying submit, bit for bit; capacities larger than the lists; several pictures through the
three rotating arenas; lists that do not sit in the arena, or exceed its capacities, are refused."""
import ctypes

import pytest

from oracle_py import Oracle
from synth_util import assert_planes_equal, make_case, oracle_decode
from test_emu_picture import emu_lib  # noqa: F401  (fixture)
from libde265_amd import capi, worklist

CASES = [dict(width=192, height=128, bit_depth=8, seed=61, n_slices=3, features=7),
         dict(width=256, height=192, bit_depth=10, seed=62, tile_cols=2, tile_rows=2, intra_pct=40, features=8 + 64 + 128 + 2),
         dict(width=128, height=128, bit_depth=8, seed=63, intra_pct=100, n_refs=0, chroma_format=3)]


def run(lib, oracle, case):
    pic, refs = make_case(**case)
    pp = pic.pp[0]
    want = oracle_decode(Oracle(oracle), pic, refs)
    ctx = capi.Context(lib, 0)
    try:
        handles = []
        for planes in refs:
            f = ctx.frame_create_for(pp)
            ctx.frame_upload(f, planes)
            handles.append(f)
        dsts = [ctx.frame_create_for(pp) for _ in range(4)]
        pic.ref_frames = [handles[i] if i < len(handles) else -1 for i in range(worklist.MAX_REF_FRAMES)]
        ctx.set_pipeline_depth(2)
        for k in range(7):                                 # three arenas: every one is used at least twice
            pic.dst_frame = dsts[k % 4]
            ctx.submit_in_place(pic, slack=1.0 + 0.4 * (k % 4), fill_threads=2)
        ctx.wait()
        for d in dsts:
            assert_planes_equal(ctx.frame_download(d), want, "in-place submit")
        # refused: lists beyond the capacities / somewhere else than the arena says
        c, keep = pic.to_c()
        caps = capi.ArenaCaps()
        for n in ("n_slices", "n_ctbs", "n_cus", "n_tus", "n_pbs", "n_wts", "n_ibs"):
            setattr(caps, n, int(getattr(c, n)) + 1)
        for b in range(4):
            caps.n_rbs[b] = int(c.rb_count[b]) + 1
        caps.n_coeffs, caps.n_pcm = int(c.n_coeffs) + 1, int(c.n_pcm) + 1
        dst = worklist.CPicture()
        ctx.L.check(ctx.L.lib.m355_arena_begin(ctx.h, ctypes.addressof(caps), ctypes.addressof(dst)))
        from libde265_amd import synth
        synth._lib().m355_synth_fill_arena(ctypes.addressof(c), ctypes.addressof(caps), ctypes.addressof(dst), 1)
        dst.dst_frame = dsts[0]
        for k in range(worklist.MAX_REF_FRAMES):
            dst.ref_frames[k] = c.ref_frames[k]
        dst.n_cus = caps.n_cus + 5
        assert ctx.L.lib.m355_submit_picture(ctx.h, ctypes.addressof(dst)) == 3          # M355_ERR_INVALID
    finally:
        ctx.close()


def run_resident(lib, oracle, case):
    """m355_picture_arena_begin on an ordinary context: lists recorded into a HANDLE's arena (new handle, then the same handle again
    with other capacities), decoded with m355_decode_resident; refusals"""
    pic, refs = make_case(**case)
    pp = pic.pp[0]
    want = oracle_decode(Oracle(oracle), pic, refs)
    ctx = capi.Context(lib, 0)
    try:
        handles = []
        for planes in refs:
            f = ctx.frame_create_for(pp)
            ctx.frame_upload(f, planes)
            handles.append(f)
        dsts = [ctx.frame_create_for(pp) for _ in range(3)]
        pic.ref_frames = [handles[i] if i < len(handles) else -1 for i in range(worklist.MAX_REF_FRAMES)]
        ctx.set_pipeline_depth(2)
        pic.dst_frame = dsts[0]
        h = ctx.upload_in_place(pic, slack=1.5, fill_threads=2)
        other = ctx.upload(pic)                                # (a handle made the copying way in between: the slots do not mix)
        assert other != h
        ctx.decode_resident(h)
        for k in (1, 2):                                       # the same handle recorded again while its last decode may still run
            pic.dst_frame = dsts[k]
            assert ctx.upload_in_place(pic, handle=h, slack=1.0 + 0.7 * (k - 1), fill_threads=2) == h
            ctx.decode_resident(h)
        ctx.wait()
        for d in dsts:
            assert_planes_equal(ctx.frame_download(d), want, "in-place resident lists")
        # refusals: a handle that does not exist; no capacities
        caps = capi.ArenaCaps()
        dst = worklist.CPicture()
        assert ctx.L.lib.m355_picture_arena_begin(ctx.h, 99, ctypes.addressof(caps), None, ctypes.addressof(dst)) == -3
        c, keep = pic.to_c()
        for n in ("n_slices", "n_ctbs", "n_cus", "n_tus", "n_pbs", "n_wts", "n_ibs"):
            setattr(caps, n, int(getattr(c, n)) + 1)
        assert ctx.L.lib.m355_picture_arena_begin(ctx.h, 99, ctypes.addressof(caps), None, ctypes.addressof(dst)) == -3
        # a tile-sharded context wants the picture parameters, and m355_arena_begin stays refused there
        ctx.release(h); ctx.release(other)
        ctx.shard_set(0, 1)
        assert ctx.L.lib.m355_picture_arena_begin(ctx.h, -1, ctypes.addressof(caps), None, ctypes.addressof(dst)) == -3
        assert b"picture parameters" in ctx.L.lib.m355_last_error()
        assert ctx.L.lib.m355_arena_begin(ctx.h, ctypes.addressof(caps), ctypes.addressof(dst)) == 3
        assert b"m355_picture_arena_begin" in ctx.L.lib.m355_last_error()
        ctx.shard_set(0, 0)
    finally:
        ctx.close()


@pytest.mark.parametrize("case", CASES, ids=lambda c: "seed%d" % c["seed"])
def test_in_place_submit_emulated(emu_lib, oracle, case):  # noqa: F811
    run(emu_lib, oracle, case)


@pytest.mark.parametrize("case", CASES[1:2], ids=lambda c: "seed%d" % c["seed"])
def test_in_place_resident_emulated(emu_lib, oracle, case):  # noqa: F811
    run_resident(emu_lib, oracle, case)


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES[:2], ids=lambda c: "seed%d" % c["seed"])
def test_in_place_resident_gpu(oracle, case):
    lib = capi.Library()
    assert lib.device_count() >= 1
    run_resident(lib, oracle, case)


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=lambda c: "seed%d" % c["seed"])
def test_in_place_submit_gpu(oracle, case):
    lib = capi.Library()
    assert lib.device_count() >= 1
    run(lib, oracle, case)
