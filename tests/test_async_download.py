"""m355_frame_download_async / m355_frame_download_wait (picture output beside later decodes) on the SIMT-interpreter build: the
planes that land equal the synchronous download, for one and for three lanes, with the frame recycled by a later picture."""
import pytest

from test_emu_picture import emu_lib  # noqa: F401  (fixture)
from synth_util import assert_planes_equal
from libde265_amd import capi, synth, worklist


@pytest.mark.parametrize("depth", [1, 3])
def test_async_download_equals_sync(emu_lib, depth):  # noqa: F811
    ctx = capi.Context(emu_lib, 0)
    try:
        ctx.set_pipeline_depth(depth)
        cfg = dict(width=128, height=64, bit_depth=8, seed=5, n_refs=1)
        pics = [synth.picture(**dict(cfg, seed=5 + k)) for k in range(4)]
        pp = pics[0].pp[0]
        r0 = ctx.frame_create_for(pp)
        ctx.frame_upload(r0, synth.ref_planes(5, 128, 64, 1, 8))
        pool = [ctx.frame_create_for(pp) for _ in range(2)]
        tokens = []
        for k, pic in enumerate(pics):
            pic.ref_frames = [r0] + [-1] * (worklist.MAX_REF_FRAMES - 1)
            pic.dst_frame = pool[k % 2]
            ctx.decode_resident(ctx.upload(pic))
            tokens.append(ctx.frame_download_async(pool[k % 2]))
        got = [ctx.frame_download_finish(t) for t in tokens]
        ctx.wait()
        # the same pictures one at a time, downloaded synchronously
        for k, pic in enumerate(pics):
            ctx.decode_resident(ctx.upload(pic))
            ctx.wait()
            assert_planes_equal(got[k], ctx.frame_download(pool[k % 2]), "picture %d" % k)
    finally:
        ctx.close()
