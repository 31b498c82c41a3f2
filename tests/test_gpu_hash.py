"""GPU parity for m355_frame_hash (SEI decoded picture hash on the device, sei.cc:161-356): MD5 / CRC / checksum of
device frames against the oracle — random pictures over all plane geometries, a decoded girlshy picture (must equal the
hashes of the reference's own planes), and an 8K 10-bit frame (size-independent property: CRC/checksum of the device
frame equal those of its download)."""
import numpy as np
import pytest

from golden_io import load_gold
from hash_util import MD5, CRC, CHECKSUM, make_planes, oracle_hash
from test_hash import GEOMS
from libde265_amd import capi

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu_lib():
    lib = capi.Library()          # raises if the HIP library is missing — no fallback
    assert lib.device_count() >= 1, "no HIP device visible"
    return lib


@pytest.mark.parametrize("geom", GEOMS + [(4096, 2160, 1, 8, 8), (7680, 4320, 1, 10, 10)])
def test_frame_hash_matches_oracle(oracle, gpu_lib, geom):
    w, h, cf, bdl, bdc = geom
    planes = make_planes(w, h, cf, bdl, bdc, seed=w + 7 * h)
    ctx = capi.Context(gpu_lib, 0)
    try:
        f = ctx.frame_create(w, h, cf, bdl, bdc)
        ctx.frame_upload(f, planes)
        bds = [bdl, bdc, bdc]
        for t in (CRC, CHECKSUM, MD5):
            assert ctx.frame_hash(f, t) == [oracle_hash(oracle, p, bds[c], t) for c, p in enumerate(planes)], "hash type %d" % t
    finally:
        ctx.close()


def test_hash_of_decoded_picture(oracle, gpu_lib):
    """decode the first girlshy pictures on the device and hash them in place: the MD5 must be the one recorded from the
    reference's planes, CRC / checksum those of the downloaded planes"""
    hdr, pics = load_gold("girlshy_full.m355gold.gz")
    ctx = capi.Context(gpu_lib, 0)
    try:
        pic = pics[0]
        dst = ctx.frame_create_for(pic.pp[0])
        saved = pic.dst_frame
        pic.dst_frame = dst
        ctx.submit(pic)
        pic.dst_frame = saved
        md5 = ctx.frame_hash(dst, MD5)        # waits for the picture
        assert [m.hex() for m in md5] == pic.meta["md5"]
        planes = ctx.frame_download(dst)
        for t in (CRC, CHECKSUM):
            assert ctx.frame_hash(dst, t) == [oracle_hash(oracle, p, 8, t) for p in planes]
    finally:
        ctx.close()
