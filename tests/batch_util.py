"""m355_decode_batch: several independent intra pictures, one k_intra launch.  Shared by the emulator tier and the GPU tier: the
planes of every picture of a batch must equal the oracle's decode of that picture alone."""
import pytest

from synth_util import assert_planes_equal, oracle_decode
from libde265_amd import capi, synth, worklist


def intra_pictures(n, **cfg):
    """n different intra pictures of one geometry (work lists from the synthetic generator, seeds apart)"""
    return [synth.picture(**dict(cfg, intra_pct=100, n_refs=0, seed=cfg["seed"] + 37 * k)) for k in range(n)]


def check_batches(lib, o, cfg, depth, batches, sizes=None, stages=worklist.STAGE_ALL):
    """decode `batches` = lists of picture indices, one m355_decode_batch each and no wait in between (frames are recycled from
    batch to batch: hazards across batches), then compare every picture's last decode"""
    n = 1 + max(max(b) for b in batches)
    pics = intra_pictures(n, **cfg)
    if sizes:                                   # pictures of different sizes in one batch: ragged work lists
        for k, (w, h) in sizes.items():
            pics[k] = synth.picture(**dict(cfg, intra_pct=100, n_refs=0, seed=cfg["seed"] + 37 * k, width=w, height=h))
    want = [oracle_decode(o, p, [], stages) for p in pics]
    ctx = capi.Context(lib, 0)
    try:
        ctx.set_pipeline_depth(depth)
        ctx.set_stages(stages)
        frames, handles = [], []
        for p in pics:
            f = ctx.frame_create_for(p.pp[0])
            p.dst_frame = f
            p.ref_frames = [-1] * worklist.MAX_REF_FRAMES
            frames.append(f)
            handles.append(ctx.upload(p))
        for b in batches:
            ctx.decode_batch([handles[k] for k in b])
        ctx.wait()
        for k in range(n):
            assert_planes_equal(ctx.frame_download(frames[k]), want[k], "picture %d" % k)
        return ctx, pics, handles, frames, want
    except BaseException:
        ctx.close()
        raise


def check_rejections(lib, o, cfg):
    """what m355_decode_batch refuses: more pictures than lanes, one picture twice, an inter picture, two pictures into one frame"""
    ctx, pics, handles, frames, want = check_batches(lib, o, cfg, 2, [[0, 1]])
    try:
        with pytest.raises(capi.M355Error):
            ctx.decode_batch([handles[0], handles[1], handles[0]])
        with pytest.raises(capi.M355Error):
            ctx.decode_batch([handles[0], handles[0]])
        inter = synth.picture(**dict(cfg, intra_pct=20, n_refs=1))
        ref = ctx.frame_create_for(inter.pp[0])
        ctx.frame_upload(ref, synth.ref_planes(5, int(inter.pp[0]["width"]), int(inter.pp[0]["height"]), int(inter.pp[0]["chroma_format_idc"]), int(inter.pp[0]["bit_depth_luma"])))
        inter.dst_frame = ctx.frame_create_for(inter.pp[0])
        inter.ref_frames = [ref] + [-1] * (worklist.MAX_REF_FRAMES - 1)
        hi = ctx.upload(inter)
        with pytest.raises(capi.M355Error):
            ctx.decode_batch([handles[0], hi])
        same = synth.picture(**dict(cfg, intra_pct=100, n_refs=0, seed=991))
        same.dst_frame = frames[0]
        same.ref_frames = [-1] * worklist.MAX_REF_FRAMES
        hs = ctx.upload(same)
        with pytest.raises(capi.M355Error):
            ctx.decode_batch([handles[0], hs])
        # the context still works afterwards, single decodes and batches mixed
        ctx.decode_resident(handles[1])
        ctx.decode_batch([handles[0], handles[1]])
        ctx.decode_resident(handles[0])
        ctx.wait()
        for k in range(2):
            assert_planes_equal(ctx.frame_download(frames[k]), want[k], "picture %d after the refused batches" % k)
    finally:
        ctx.close()
