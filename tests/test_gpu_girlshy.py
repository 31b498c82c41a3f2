"""GPU parity (config C1): the HIP kernels, driven through the C ABI, must reproduce every plane of
every picture the REAL reference decoder produced for testdata/girlshy.h265 (75 pictures: intra,
P/B inter with weighted prediction, deblocking, SAO) and hence the reference CI's golden MD5
b81538fa33a67278e5263e231e43ca98 (scripts/ci-run.sh:91-92).  Work lists and expected plane hashes
come from tests/golden/ (recorded from the reference by tests/golden/make_girlshy_fixture.py)."""
import hashlib

import pytest

from golden_io import load_gold
from oracle_py import plane_md5s
from test_emu_picture import run_stream
from test_girlshy_oracle import replay
from libde265_amd import capi, worklist

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu_lib():
    lib = capi.Library()          # raises if the HIP library is missing — no fallback
    assert lib.device_count() >= 1, "no HIP device visible"
    return lib


@pytest.mark.parametrize("variant", ["full", "nolf", "nosao", "nodeblk"])
def test_girlshy_bit_exact_on_gpu(gpu_lib, variant):
    hdr, pics = load_gold("girlshy_%s.m355gold.gz" % variant)
    ctx = capi.Context(gpu_lib, 0)
    try:
        run_stream(ctx, pics, len(pics))
    finally:
        ctx.close()


def test_girlshy_stream_md5_on_gpu(gpu_lib):
    """whole-stream MD5 as `dec265 -o -` writes it (display order, conformance-window crop)"""
    hdr, pics = load_gold("girlshy_full.m355gold.gz")
    ctx = capi.Context(gpu_lib, 0)
    by_poc = {}

    def decode(pic, dst, refs):
        pic.dst_frame = dst
        saved = pic.ref_frames
        pic.ref_frames = [refs.get(s, -1) for s in range(worklist.MAX_REF_FRAMES)]
        ctx.submit(pic)
        pic.ref_frames = saved
        return 0

    try:
        dpb = [p.dst_frame for p in pics]
        for i, pic, planes in replay(decode, ctx.frame_create_for, ctx.frame_destroy, ctx.frame_download, pics):
            pic.dst_frame = dpb[i]
            by_poc[pic.meta["poc"]] = planes
        ctx.wait()
    finally:
        ctx.close()
    m = hashlib.md5()
    for poc, w, h, cx, cy in hdr["order"]:
        pl = by_poc[poc]
        for c, p in enumerate(pl):
            sx = 1 if c == 0 else pl[0].shape[1] // p.shape[1]
            sy = 1 if c == 0 else pl[0].shape[0] // p.shape[0]
            m.update(p[cy // sy:cy // sy + h // sy, cx // sx:cx // sx + w // sx].tobytes())
    assert m.hexdigest() == "b81538fa33a67278e5263e231e43ca98"


@pytest.mark.parametrize("depth", [2, 3, 4])
def test_girlshy_pipelined(gpu_lib, depth):
    """the real stream's dependency structure (I/P/B reference chains) with several pictures in flight: all 75 pictures are
    submitted back to back (one frame each, no host synchronisation in between), then compared with the reference"""
    hdr, pics = load_gold("girlshy_full.m355gold.gz")
    ctx = capi.Context(gpu_lib, 0)
    try:
        ctx.set_pipeline_depth(depth)
        frames, by_dpb = [], {}
        for pic in pics:
            dst = ctx.frame_create_for(pic.pp[0])
            saved = (pic.dst_frame, pic.ref_frames)
            pic.ref_frames = [by_dpb[s] if pic.ref_frames[s] >= 0 else -1 for s in range(worklist.MAX_REF_FRAMES)]
            by_dpb[pic.dst_frame] = dst
            pic.dst_frame = dst
            ctx.submit(pic)
            pic.dst_frame, pic.ref_frames = saved
            frames.append(dst)
        ctx.wait()
        for i, (pic, f) in enumerate(zip(pics, frames)):
            assert plane_md5s(ctx.frame_download(f)) == pic.meta["md5"], "picture %d (POC %d) differs from the reference" % (i, pic.meta["poc"])
    finally:
        ctx.close()
