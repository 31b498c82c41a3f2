"""Synthetic HEVC bitstreams (oracle/ref_streamgen.cc: our coding-tree syntax writer on the reference's own CABAC encoder,
parameter-set / slice-header writers and residual writer) decoded LIVE through de265.h twice — by the reference decoder
(oracle/_ref/libde265_ref.so, scalar and SIMD) and by glue/_build/libde265.so, whose pixels come from the MI355X backend —
and compared bit for bit.  This is what pins the glue's recorder and the kernels on what girlshy does not contain: P and B
slices with merge / skip / AMVP motion the DECODER derives, two reference pictures, uniform tiles decoded by parallel threads,
10-bit samples, AMP partitions, intra NxN inside inter pictures, SAO parameters with merge candidates.

CPU tier: small streams, backend = SIMT-interpreter build.  GPU tier: up to 4K tiled 10-bit."""
import ctypes
import hashlib
import os
import subprocess

import pytest

import de265_py
from libde265_amd import capi
from test_emu_picture import emu_lib, EMU_SO  # noqa: F401  (fixture)
from test_glue_live import glue_lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STREAMGEN = os.path.join(ROOT, "oracle", "_ref", "streamgen")


def make_stream(tmp_path, w, h, bd, tc, tr, frames, seed, intra_pct=5, b_frames=1, sao=1):
    if not os.path.exists(STREAMGEN):
        if not os.path.isdir("/root/reference"):
            pytest.skip("oracle/_ref/streamgen not available here")
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "-j8", "gen"], check=True, stdout=subprocess.DEVNULL)
    out = os.path.join(str(tmp_path), "s_%dx%d_%d_%dx%d_%d.h265" % (w, h, bd, tc, tr, seed))
    subprocess.run([STREAMGEN, out, str(w), str(h), str(bd), str(tc), str(tr), str(frames), str(seed), str(intra_pct), str(b_frames), str(sao)], check=True)
    return open(out, "rb").read()


def check(ref, data, frames, threads, backend):
    want = de265_py.decode_stream(ref, data, threads=0, scalar=True)
    assert want[1] == frames and not want[2], "the reference itself rejects the generated stream: %r" % (want,)
    assert de265_py.decode_stream(ref, data, threads=threads)[0] == want[0]          # SIMD tables, tile threads
    lib = glue_lib()
    got = de265_py.decode_stream(lib, data, threads=threads)
    assert got[:2] == want[:2], "live decode on the backend differs from the reference decoder"
    assert set(got[2]) <= {1000}, got[2]       # (DE265_WARNING_NO_WPP_CANNOT_USE_MULTITHREADING: threads asked for on a stream without tiles)
    assert lib.m355_glue_cpu_pixel_calls() == 0
    assert os.path.realpath(lib.m355_glue_backend_path().decode()) == os.path.realpath(backend)


CPU_CASES = [(416, 240, 8, 1, 1, 4, 21), (448, 256, 10, 2, 2, 4, 22)]


@pytest.mark.parametrize("w,h,bd,tc,tr,frames,seed", CPU_CASES)
def test_generated_streams_emulated_backend(ref, emu_lib, tmp_path, monkeypatch, w, h, bd, tc, tr, frames, seed):  # noqa: F811
    monkeypatch.setenv("M355_LIB", EMU_SO)
    check(ref, make_stream(tmp_path, w, h, bd, tc, tr, frames, seed), frames, 4, EMU_SO)


GPU_CASES = [(416, 240, 8, 1, 1, 8, 31, 5, 1, 1), (832, 480, 10, 3, 2, 8, 32, 10, 1, 1), (1920, 1080, 8, 2, 1, 5, 33, 30, 0, 1),
             (1280, 720, 12, 2, 2, 5, 34, 5, 1, 0), (3840, 2160, 10, 2, 2, 3, 35, 5, 1, 1), (640, 368, 8, 1, 1, 6, 36, 100, 0, 1)]


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,bd,tc,tr,frames,seed,intra,b,sao", GPU_CASES)
def test_generated_streams_gpu(ref, tmp_path, monkeypatch, w, h, bd, tc, tr, frames, seed, intra, b, sao):
    monkeypatch.delenv("M355_LIB", raising=False)
    check(ref, make_stream(tmp_path, w, h, bd, tc, tr, frames, seed, intra, b, sao), frames, 8, capi.DEFAULT_LIB)
