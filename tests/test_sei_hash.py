"""Decoded-picture-hash SEIs (what conformance streams carry, `dec265 -c`) checked against the DEVICE frame: the glue replaces
process_sei (sei.cc:436) by m355_frame_hash — CRC and checksum by reduction kernels on the device, MD5 by the backend on its own
copy — so a stream with hash SEIs never downloads a picture into the application's planes.  The SEIs are written by
tests/sei_util.py from the REFERENCE decoder's output and first accepted by the reference decoder itself; a stream whose last
picture carries a wrong hash must be refused by both (DE265_ERROR_CHECKSUM_MISMATCH)."""
import ctypes

import pytest

import de265_py
import sei_util
from libde265_amd import capi
from test_emu_picture import emu_lib, EMU_SO  # noqa: F401  (fixture)
from test_glue_live import glue_lib
from test_streams import make_stream, F_REXT, F_TSKIP, F_WP


def run(ref, oracle, tmp_path, w, h, bd, tc, tr, frames, seed, feat, chroma, hash_type, threads):
    data = make_stream(tmp_path, w, h, bd, tc, tr, frames, seed, 10, 1, 1, feat, chroma, 1)
    pics = []
    plain = de265_py.decode_stream(ref, data, planes_out=pics)
    assert plain[1] == frames and not plain[2]
    good = sei_util.add_hash_seis(data, pics, bd, hash_type, oracle)
    bad = sei_util.add_hash_seis(data, pics, bd, hash_type, oracle, corrupt_picture=frames - 1)
    # the reference's own verdict on our SEIs (not for the checksum of > 8-bit pictures: its compute_checksum halves a stride that
    # is already in samples, sei.cc:173-174, and refuses every such picture — see tests/test_hash.py, oracle/hevc_oracle.c:o_hash_checksum)
    if not (hash_type == sei_util.CHECKSUM and bd > 8):
        assert de265_py.decode_stream(ref, good, check_hash=True)[1:] == (frames, [])
        assert de265_py.decode_stream(ref, bad, check_hash=True)[2] == [5]          # DE265_ERROR_CHECKSUM_MISMATCH
    lib = glue_lib()
    lib.m355_glue_hashed_pictures.restype = ctypes.c_longlong
    lib.m355_glue_hashed_pictures.argtypes = [ctypes.c_void_p]
    lib.m355_glue_stats.argtypes = [ctypes.c_void_p] + [ctypes.POINTER(ctypes.c_longlong)] * 3
    seen = {}

    def stats(ctx):
        n = [ctypes.c_longlong() for _ in range(3)]
        assert lib.m355_glue_stats(ctx, *[ctypes.byref(x) for x in n]) == 0
        seen.update(pictures=n[0].value, downloads=n[2].value, hashed=lib.m355_glue_hashed_pictures(ctx))
    got = de265_py.decode_stream(lib, good, threads=threads, check_hash=True, touch_planes=False, before_free=stats)
    assert got[1] == frames and set(got[2]) <= {1000}, got
    assert seen == dict(pictures=frames, downloads=0, hashed=frames), seen      # every picture hashed where it lives, none brought back
    assert lib.m355_glue_cpu_pixel_calls() == 0
    assert 5 in de265_py.decode_stream(lib, bad, threads=threads, check_hash=True, touch_planes=False)[2]
    # and with the application taking the pictures as well: same output as the reference
    assert de265_py.decode_stream(lib, good, threads=threads, check_hash=True)[0] == plain[0]


@pytest.mark.parametrize("hash_type", [sei_util.MD5, sei_util.CRC, sei_util.CHECKSUM])
def test_sei_hash_emulated_backend(ref, oracle, emu_lib, tmp_path, monkeypatch, hash_type):  # noqa: F811
    monkeypatch.setenv("M355_LIB", EMU_SO)
    run(ref, oracle, tmp_path, 192, 128, 8, 1, 1, 3, 81 + hash_type, 0, 1, hash_type, 2)


GPU_CASES = [(832, 480, 8, 1, 1, 6, 91, 0, 1), (1280, 720, 10, 2, 2, 5, 92, F_WP, 1), (832, 480, 10, 1, 1, 4, 93, F_REXT | F_TSKIP, 3), (832, 480, 8, 2, 1, 4, 94, 0, 0)]


@pytest.mark.gpu
@pytest.mark.parametrize("hash_type", [sei_util.MD5, sei_util.CRC, sei_util.CHECKSUM])
@pytest.mark.parametrize("w,h,bd,tc,tr,frames,seed,feat,chroma", GPU_CASES)
def test_sei_hash_gpu(ref, oracle, tmp_path, monkeypatch, w, h, bd, tc, tr, frames, seed, feat, chroma, hash_type):
    monkeypatch.delenv("M355_LIB", raising=False)
    run(ref, oracle, tmp_path, w, h, bd, tc, tr, frames, seed, feat, chroma, hash_type, 8)
