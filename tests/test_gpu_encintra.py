"""GPU parity on the reference-encoder intra streams (see test_encintra_streams.py): every plane of every picture as the
reference decoder produced it, and the MD5 of the YUV file the reference CLI writes for the stream."""
import pytest

from golden_io import load_gold
from oracle_py import plane_md5s
from test_encintra_streams import STREAMS, stream_md5
from libde265_amd import capi

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu_lib():
    lib = capi.Library()          # raises if the HIP library is missing — no fallback
    assert lib.device_count() >= 1, "no HIP device visible"
    return lib


@pytest.mark.parametrize("depth", [1, 3])
@pytest.mark.parametrize("name", sorted(STREAMS))
def test_encintra_bit_exact_on_gpu(gpu_lib, name, depth):
    hdr, pics = load_gold("encintra_%s.m355gold.gz" % name)
    ctx = capi.Context(gpu_lib, 0)
    try:
        ctx.set_pipeline_depth(depth)
        frames = []
        for pic in pics:                      # IDR pictures: no references; all submitted back to back
            saved = pic.dst_frame
            pic.dst_frame = ctx.frame_create_for(pic.pp[0])
            ctx.submit(pic)
            frames.append(pic.dst_frame)
            pic.dst_frame = saved
        ctx.wait()
        out = [ctx.frame_download(f) for f in frames]
        for i, (pic, pl) in enumerate(zip(pics, out)):
            assert plane_md5s(pl) == pic.meta["md5"], "picture %d differs from the reference" % i
        assert stream_md5(hdr, out, pics) == STREAMS[name]
    finally:
        ctx.close()
