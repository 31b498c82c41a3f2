"""Two pictures in flight (m355_set_pipeline_depth(ctx, 2)): consecutive decodes alternate between two lanes and
overlap; frame hazards are ordered by events.  A chain of pictures in which every picture references the PREVIOUS
output (read-after-write across lanes) and frames are recycled (write-after-read / write-after-write) must still
equal the oracle's sequential decode, with no host synchronisation between the submissions."""
import numpy as np
import pytest

from oracle_py import Oracle
from synth_util import assert_planes_equal
from libde265_amd import capi, synth, worklist

pytestmark = pytest.mark.gpu


def chain_case(n, **cfg):
    pics = [synth.picture(**dict(cfg, seed=cfg["seed"] + 101 * k)) for k in range(n)]
    pp = pics[0].pp[0]
    ref0 = synth.ref_planes(cfg["seed"], int(pp["width"]), int(pp["height"]), int(pp["chroma_format_idc"]), int(pp["bit_depth_luma"]))
    return pics, ref0


@pytest.mark.parametrize("cfg", [dict(width=416, height=240, bit_depth=8, seed=201, n_refs=2),
                                 dict(width=832, height=480, bit_depth=10, seed=202, n_refs=2, tile_cols=2, tile_rows=2),
                                 dict(width=640, height=368, bit_depth=8, seed=203, n_refs=2, sao=0)],
                         ids=lambda c: "%dx%d_seed%d" % (c["width"], c["height"], c["seed"]))
@pytest.mark.parametrize("depth", [2, 3, 4])
def test_dependent_pictures_pipelined(oracle, cfg, depth):
    o = Oracle(oracle)
    n = 6
    pics, ref0 = chain_case(n, **cfg)
    pp = pics[0].pp[0]
    # oracle: picture k references slot 0 = the fixed frame, slot 1 = output of picture k-1 (k = 0: the fixed frame again)
    of0 = o.frame_new(pp); o.frame_set_planes(of0, ref0)
    oprev, want = of0, []
    for pic in pics:
        od = o.frame_new(pp)
        pic.ref_frames = [0, 1] + [-1] * (worklist.MAX_REF_FRAMES - 2)
        assert o.decode(pic, od, {0: of0, 1: oprev}) == 0
        want.append(o.frame_planes(od))
        oprev = od
    lib = capi.Library()
    ctx = capi.Context(lib, 0)
    try:
        ctx.set_pipeline_depth(depth)
        g0 = ctx.frame_create_for(pp); ctx.frame_upload(g0, ref0)
        pool = [ctx.frame_create_for(pp) for _ in range(3)]        # recycled destination frames
        handles, prev = [], g0
        for k, pic in enumerate(pics):
            pic.dst_frame = pool[k % 3]
            pic.ref_frames = [g0, prev] + [-1] * (worklist.MAX_REF_FRAMES - 2)
            handles.append(ctx.upload(pic))
            prev = pic.dst_frame
        got = []
        # pictures 0..2 go out back to back; their frames are then recycled by 3..5, so download in between
        for k in range(3):
            ctx.decode_resident(handles[k])
        ctx.wait()
        got += [ctx.frame_download(pool[k]) for k in range(3)]
        for k in range(3, n):
            ctx.decode_resident(handles[k])
        ctx.wait()
        got += [ctx.frame_download(pool[k % 3]) for k in range(3, n)]
        for k in range(n):
            assert_planes_equal(got[k], want[k], "picture %d" % k)
        # replaying one resident picture many times (the benchmark's pattern: same destination every time)
        for _ in range(8):
            ctx.decode_resident(handles[n - 1])
        ctx.wait()
        assert_planes_equal(ctx.frame_download(pool[(n - 1) % 3]), want[n - 1], "replayed")
    finally:
        ctx.close()


@pytest.mark.parametrize("depth", [1, 3])
def test_async_download_beside_later_decodes(oracle, depth):
    """m355_frame_download_async: the copy of picture k is started right behind its decode and runs beside the decodes of k+1, k+2
    (which reference k and then RECYCLE its frame: the writer must wait for the copy); what lands in the pinned planes is picture k."""
    o = Oracle(oracle)
    cfg = dict(width=832, height=480, bit_depth=10, seed=311, n_refs=2, tile_cols=2, tile_rows=1)
    n = 7
    pics, ref0 = chain_case(n, **cfg)
    pp = pics[0].pp[0]
    of0 = o.frame_new(pp); o.frame_set_planes(of0, ref0)
    oprev, want = of0, []
    for pic in pics:
        od = o.frame_new(pp)
        pic.ref_frames = [0, 1] + [-1] * (worklist.MAX_REF_FRAMES - 2)
        assert o.decode(pic, od, {0: of0, 1: oprev}) == 0
        want.append(o.frame_planes(od))
        oprev = od
    lib = capi.Library()
    ctx = capi.Context(lib, 0)
    try:
        ctx.set_pipeline_depth(depth)
        g0 = ctx.frame_create_for(pp); ctx.frame_upload(g0, ref0)
        pool = [ctx.frame_create_for(pp) for _ in range(2)]        # two destination frames: picture k+2 overwrites picture k's
        handles, prev = [], g0
        for k, pic in enumerate(pics):
            pic.dst_frame = pool[k % 2]
            pic.ref_frames = [g0, prev] + [-1] * (worklist.MAX_REF_FRAMES - 2)
            handles.append(ctx.upload(pic))
            prev = pic.dst_frame
        tokens = []
        for k in range(n):                                          # no host synchronisation anywhere in this loop
            ctx.decode_resident(handles[k])
            tokens.append(ctx.frame_download_async(pool[k % 2]))
        def check(k, what):
            got = ctx.frame_download_finish(tokens[k])
            tokens[k] = None
            same_as = [j for j in range(n) if all(np.array_equal(a, b) for a, b in zip(got, want[j]))]
            assert same_as == [k], "%s: the planes of download %d hold picture %s" % (what, k, same_as or "nothing the oracle made")
        for k in (n - 1, 0, 3):                                     # any order
            check(k, "async, out of order")
        ctx.wait()                                                  # m355_wait covers the copies too
        for k in range(n):
            if tokens[k] is not None:
                check(k, "async")
    finally:
        ctx.close()


@pytest.mark.parametrize("depth", [4, 9])
def test_intra_and_inter_pictures_alternate_on_deep_pipelines(oracle, depth):
    """Lanes 3.. decode intra pictures on a stream of another priority class (own hardware queues) and inter pictures on the lane's
    ordinary streams: a lane's scratch is shared by both, frames are recycled across lanes, inter pictures reference the picture
    decoded right before them (an intra one, on another stream) — everything must still equal the oracle's sequential decode."""
    o = Oracle(oracle)
    base = dict(width=416, height=240, bit_depth=8, n_refs=1)
    n = 2 * depth + 3
    pics = []
    for k in range(n):
        intra = (k % 2 == 0) or (k % 5 == 0)
        pics.append(synth.picture(**dict(base, seed=500 + 7 * k, intra_pct=100 if intra else 15, n_refs=0 if intra else 1)))
    pp = pics[0].pp[0]
    ref0 = synth.ref_planes(77, int(pp["width"]), int(pp["height"]), int(pp["chroma_format_idc"]), int(pp["bit_depth_luma"]))
    of0 = o.frame_new(pp); o.frame_set_planes(of0, ref0)
    oprev, want = of0, []
    for pic in pics:
        od = o.frame_new(pp)
        pic.ref_frames = [0] + [-1] * (worklist.MAX_REF_FRAMES - 1)
        assert o.decode(pic, od, {0: oprev}) == 0
        want.append(o.frame_planes(od))
        oprev = od
    lib = capi.Library()
    ctx = capi.Context(lib, 0)
    try:
        ctx.set_pipeline_depth(depth)
        g0 = ctx.frame_create_for(pp); ctx.frame_upload(g0, ref0)
        npool = 4
        pool = [ctx.frame_create_for(pp) for _ in range(npool)]
        handles, prev = [], g0
        for k, pic in enumerate(pics):
            pic.dst_frame = pool[k % npool]
            pic.ref_frames = [prev] + [-1] * (worklist.MAX_REF_FRAMES - 1)
            handles.append(ctx.upload(pic))
            prev = pic.dst_frame
        got = []
        for k0 in range(0, n, npool - 1):          # frames are recycled: take the finished ones out before their next writer is issued
            ks = list(range(k0, min(n, k0 + npool - 1)))
            for k in ks:
                ctx.decode_resident(handles[k])
            ctx.wait()
            got += [ctx.frame_download(pool[k % npool]) for k in ks]
        for k in range(n):
            assert_planes_equal(got[k], want[k], "picture %d" % k)
        # and without any host synchronisation in between: the last npool pictures are still in their frames afterwards
        for _ in range(3):
            for k in range(n):
                ctx.decode_resident(handles[k])
        ctx.wait()
        for k in range(n - npool + 1, n):
            assert_planes_equal(ctx.frame_download(pool[k % npool]), want[k], "picture %d after three unsynchronised rounds" % k)
    finally:
        ctx.close()


def long_chain(lib, oracle, cfg, depth, n_decodes, n_pool=4):
    """More decodes than the context's ring of completion marks holds (runtime.hip EvRef: 256), none of them waited for by the host:
    every picture references the output of the one before it (read-after-write across lanes), four destination frames go round
    (write-after-read / write-after-write against marks that have long been taken over), six lists go round their handles."""
    o = Oracle(oracle)
    pics, ref0 = chain_case(6, **cfg)
    pp = pics[0].pp[0]
    of0 = o.frame_new(pp); o.frame_set_planes(of0, ref0)
    opool = [o.frame_new(pp) for _ in range(n_pool)]
    oprev = of0
    for k in range(n_decodes):
        pic = pics[k % 6]
        pic.ref_frames = [0, 1] + [-1] * (worklist.MAX_REF_FRAMES - 2)
        assert o.decode(pic, opool[k % n_pool], {0: of0, 1: oprev}) == 0
        oprev = opool[k % n_pool]
    want = [o.frame_planes(f) for f in opool]
    ctx = capi.Context(lib, 0)
    try:
        ctx.set_pipeline_depth(depth)
        g0 = ctx.frame_create_for(pp); ctx.frame_upload(g0, ref0)
        pool = [ctx.frame_create_for(pp) for _ in range(n_pool)]
        handles = [None] * 6
        prev = g0
        for k in range(n_decodes):
            pic = pics[k % 6]
            pic.dst_frame = pool[k % n_pool]
            pic.ref_frames = [g0, prev] + [-1] * (worklist.MAX_REF_FRAMES - 2)
            # the lists go back into their handle with the new frames (m355_picture_replace waits for that handle's last decode only)
            if handles[k % 6] is None:
                handles[k % 6] = ctx.upload(pic)
            else:
                c, keep = pic.to_c()
                import ctypes
                ctx.L.check(ctx.L.lib.m355_picture_replace(ctx.h, handles[k % 6], ctypes.addressof(c)))
            ctx.decode_resident(handles[k % 6])
            prev = pic.dst_frame
        ctx.wait()
        for k in range(n_pool):
            assert_planes_equal(ctx.frame_download(pool[k]), want[k], "frame %d after %d decodes" % (k, n_decodes))
    finally:
        ctx.close()


@pytest.mark.parametrize("depth", [3, 4])
def test_long_unsynchronised_chain(oracle, depth):
    long_chain(capi.Library(), oracle, dict(width=416, height=240, bit_depth=8, seed=211, n_refs=2), depth, 700)


@pytest.mark.parametrize("sao", [0, 1])
@pytest.mark.parametrize("depth", [2, 3, 5])
def test_chain_whose_destination_the_previous_picture_still_reads(oracle, depth, sao):
    """Two destination frames go round: picture k writes the frame picture k - 1 reads as its reference (write-after-read at distance 1) and
    reads the one picture k - 1 writes.  A chain picture's front part runs on a lane of its own beside its reference's last stages and its
    back part on the reference's stream (runtime_decode.hip); without SAO the destination's hazards are waited for in front of k_inter, behind
    that change of stream."""
    long_chain(capi.Library(), oracle, dict(width=640, height=368, bit_depth=8, seed=213 + sao, n_refs=2, sao=sao), depth, 300, n_pool=2)
