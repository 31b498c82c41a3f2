"""m355_frame_fill: the device form of the "unavailable reference picture" the reference generates when a stream loses a
reference (decctx.cc:1294-1321 -> de265_image::fill_image with 1 << (BitDepth - 1)).  A picture predicted from a FILLED frame
must equal the oracle's picture predicted from constant planes — on the SIMT interpreter (CPU tier) and on the GPU."""
import numpy as np
import pytest

from oracle_py import Oracle
from synth_util import assert_planes_equal, make_case, oracle_decode
from test_emu_picture import emu_lib  # noqa: F401  (fixture)
from libde265_amd import capi, worklist

CASES = [dict(width=192, height=128, bit_depth=8, seed=51, intra_pct=5),
         dict(width=192, height=128, bit_depth=10, seed=52, intra_pct=5, chroma_format=3, weighted_pct=40),
         dict(width=128, height=64, bit_depth=12, seed=53, intra_pct=0, chroma_format=4, features=256)]


def run(lib, oracle, case):
    pic, refs = make_case(**case)
    pp = pic.pp[0]
    bd = int(pp["bit_depth_luma"])
    vl, vc = 1 << (bd - 1), (1 << (bd - 1)) - 3          # distinct luma / chroma values: a swapped argument would show
    refs = [[np.full_like(p, vl if c == 0 else vc) for c, p in enumerate(planes)] for planes in refs]
    want = oracle_decode(Oracle(oracle), pic, refs)
    ctx = capi.Context(lib, 0)
    try:
        handles = []
        for _ in refs:
            f = ctx.frame_create_for(pp)
            ctx.frame_fill(f, vl, vc)
            handles.append(f)
        # the fill itself
        for c, p in enumerate(ctx.frame_download(handles[0])):
            assert np.all(p == (vl if c == 0 else vc))
        pic.dst_frame = ctx.frame_create_for(pp)
        pic.ref_frames = [handles[i] if i < len(handles) else -1 for i in range(worklist.MAX_REF_FRAMES)]
        ctx.submit(pic)
        ctx.wait()
        assert_planes_equal(ctx.frame_download(pic.dst_frame), want, "predicted from filled frames")
    finally:
        ctx.close()


@pytest.mark.parametrize("case", CASES, ids=lambda c: "seed%d" % c["seed"])
def test_frame_fill_emulated(emu_lib, oracle, case):  # noqa: F811
    run(emu_lib, oracle, case)


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=lambda c: "seed%d" % c["seed"])
def test_frame_fill_gpu(oracle, case):
    lib = capi.Library()
    assert lib.device_count() >= 1
    run(lib, oracle, case)
