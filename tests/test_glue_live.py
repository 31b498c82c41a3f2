"""LIVE decode through the reference's public API (de265.h) with every pixel produced by the MI355X backend.

glue/_build/libde265.so = the reference's own objects + glue/m355_glue.cc (see its header): the parser records work lists
instead of calling the pixel functions, each complete picture goes to m355_submit_picture, pictures stay on the device keyed
by DPB index and are downloaded into (pinned) host planes when the application takes them.  The tests drive that library
like dec265 does and require the reference's golden MD5s (scripts/ci-run.sh:91-92 for the full chain; the three
stage-isolated ones of SURVEY 8c) with ZERO calls into the decoder's CPU pixel table (every slot is a counting trap).

CPU tier: the same library with the backend swapped (M355_LIB) for the SIMT-interpreter build of the product kernels.
GPU tier: the product library on the device."""
import ctypes
import os
import subprocess

import pytest

import de265_py
from libde265_amd import capi
from test_emu_picture import emu_lib, EMU_SO  # noqa: F401  (fixture)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GLUE_DIR = os.path.join(ROOT, "glue")
GLUE_SO = os.path.join(GLUE_DIR, "_build", "libde265.so")
STREAM = os.path.join(GLUE_DIR, "_build", "testdata", "girlshy.h265")
MD5 = {"full": "b81538fa33a67278e5263e231e43ca98", "nosao": "f0647c472db1a58c4a5f8b605da7513b",
       "nodeblk": "6983e435cf17979b90b721555a45493b", "nolf": "098a8f4d62bef69504174073879cd4ad"}
VARIANT = {"full": {}, "nosao": dict(disable_sao=True), "nodeblk": dict(disable_deblocking=True),
           "nolf": dict(disable_sao=True, disable_deblocking=True)}


def glue_lib():
    if not (os.path.exists(GLUE_SO) and os.path.exists(STREAM)):
        if not os.path.isdir("/root/reference"):
            pytest.skip("glue build (glue/_build) not available here")
        subprocess.run(["make", "-s", "-C", GLUE_DIR, "-j8"], check=True, stdout=subprocess.DEVNULL)
    lib = ctypes.CDLL(GLUE_SO)
    lib.m355_glue_cpu_pixel_calls.restype = ctypes.c_longlong
    lib.m355_glue_backend_path.restype = ctypes.c_char_p
    return lib


def run(variant, threads, scalar=False, backend=None, prefetch=True):
    lib = glue_lib()
    lib.m355_glue_prefetched_pictures.restype = ctypes.c_longlong
    lib.m355_glue_prefetched_pictures.argtypes = [ctypes.c_void_p]
    stats = {}

    def grab(ctx):
        stats["ctx"] = ctx

    def count(ctx):
        stats["prefetched"] = lib.m355_glue_prefetched_pictures(ctx)

    data = open(STREAM, "rb").read()
    md5, n, warnings = de265_py.decode_stream(lib, data, threads=threads, scalar=scalar, after_create=grab, before_free=count, **VARIANT[variant])
    # the application takes every picture: once the first one has been read, the downloads of the pictures submitted after that are started behind their decodes (unless switched off)
    assert (stats["prefetched"] > 0) if prefetch else (stats["prefetched"] == 0), stats
    assert n == 75 and not warnings
    assert md5 == MD5[variant], "live decode (%s, %d threads) differs from the reference's golden MD5" % (variant, threads)
    assert lib.m355_glue_cpu_pixel_calls() == 0, "the decoder called into its CPU pixel table"
    # the backend is chosen once per process (M355_LIB at the first decoder): make sure this run used the intended one
    assert os.path.realpath(lib.m355_glue_backend_path().decode()) == os.path.realpath(backend), lib.m355_glue_backend_path()


# ---- CPU tier: backend = SIMT-interpreter build of the product kernels (checks the glue, the recorder and the runtime) ----
@pytest.mark.parametrize("variant,threads", [("full", 4), ("nolf", 0)])
def test_live_decode_emulated_backend(emu_lib, variant, threads, monkeypatch):  # noqa: F811
    monkeypatch.setenv("M355_LIB", EMU_SO)
    run(variant, threads, backend=EMU_SO)


def test_live_decode_emulated_backend_without_download_prefetch(emu_lib, monkeypatch):  # noqa: F811
    monkeypatch.setenv("M355_LIB", EMU_SO)
    monkeypatch.setenv("M355_GLUE_NO_PREFETCH", "1")
    run("nolf", 2, backend=EMU_SO, prefetch=False)


# ---- GPU tier ----
@pytest.mark.gpu
@pytest.mark.parametrize("variant,threads", [("full", 0), ("full", 8), ("nosao", 2), ("nodeblk", 0), ("nolf", 3)])
def test_live_decode_gpu(variant, threads, monkeypatch):
    monkeypatch.delenv("M355_LIB", raising=False)
    run(variant, threads, backend=capi.DEFAULT_LIB)


@pytest.mark.gpu
def test_live_decode_gpu_traps_survive_acceleration_switch(monkeypatch):
    """dec265 -0 (DE265_DECODER_PARAM_ACCELERATION_CODE = SCALAR) refills the decoder's table; the glue re-arms its traps"""
    monkeypatch.delenv("M355_LIB", raising=False)
    run("full", 2, scalar=True, backend=capi.DEFAULT_LIB)
