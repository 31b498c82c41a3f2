"""Second family of real-bitstream pins: intra streams produced by the reference's own encoder (tests/golden/
make_enc_fixture.py, oracle/ref_encode.cc), decoded by the reference decoder and recorded like girlshy.  They cover what
girlshy does not: CTB sizes 16 and 32, pictures ending in partial CTBs, dense high-quality residuals (QP 10), every
intra mode on real encoder decisions.  CPU tier: the oracle and the emulated product kernels; the GPU tier replays the
same fixtures on the device (test_gpu_encintra.py)."""
import hashlib

import pytest

from golden_io import load_gold
from oracle_py import Oracle, plane_md5s
from test_girlshy_oracle import replay
from test_emu_picture import emu_lib, run_stream  # noqa: F401  (fixture)
from libde265_amd import capi

STREAMS = {"ctb64": "e70d13b4832ca5890f36f626b917b51a", "ctb32_hq": "43aa58a2f180812587388e53eb12e1b4",
           "ctb16_lq": "ef311f22cfda38d827615120e4276109"}


def stream_md5(hdr, planes_in_decode_order, pics):
    """MD5 of the YUV file `dec265 -o` writes: cropped planes in output order.  IDR-only streams repeat POC 0, so
    pictures of one POC are output in the order they were decoded."""
    by_poc = {}
    for pic, pl in zip(pics, planes_in_decode_order):
        by_poc.setdefault(pic.meta["poc"], []).append(pl)
    m = hashlib.md5()
    for poc, w, h, cx, cy in hdr["order"]:
        pl = by_poc[poc].pop(0)
        for c, p in enumerate(pl):
            sx = 1 if c == 0 else pl[0].shape[1] // p.shape[1]
            sy = 1 if c == 0 else pl[0].shape[0] // p.shape[0]
            m.update(p[cy // sy:cy // sy + h // sy, cx // sx:cx // sx + w // sx].tobytes())
    return m.hexdigest()


@pytest.mark.parametrize("name", sorted(STREAMS))
def test_oracle_matches_reference(oracle, name):
    hdr, pics = load_gold("encintra_%s.m355gold.gz" % name)
    assert hdr["stream_md5"] == STREAMS[name]
    o = Oracle(oracle)
    out = []
    for i, pic, planes in replay(o.decode, o.frame_new, o.frame_free, o.frame_planes, pics):
        assert plane_md5s(planes) == pic.meta["md5"], "picture %d differs from the reference" % i
        out.append([p.copy() for p in planes])
    assert stream_md5(hdr, out, pics) == STREAMS[name]


@pytest.mark.parametrize("name", sorted(STREAMS))
def test_emulated_kernels_match_reference(emu_lib, name):  # noqa: F811
    hdr, pics = load_gold("encintra_%s.m355gold.gz" % name)
    ctx = capi.Context(emu_lib, 0)
    try:
        run_stream(ctx, pics, len(pics))
    finally:
        ctx.close()
