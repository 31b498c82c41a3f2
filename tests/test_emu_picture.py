"""Kernel-logic verification (CPU tier): the product's .hip kernels compiled UNCHANGED under the SIMT
interpreter in tests/simt_emu/ (see its header for what that is) must reproduce the reference on the
recorded girlshy pictures.  This exercises the same C ABI the GPU tier uses; it checks indexing, LDS
staging, barrier placement and the wavefront protocol's bookkeeping — not the hardware memory model
and not performance.  The GPU tier (tests/test_gpu_*.py) is the parity gate proper."""
import os
import subprocess

import pytest

from golden_io import load_gold
from oracle_py import plane_md5s
from test_girlshy_oracle import replay
from libde265_amd import capi, worklist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU_SO = os.path.join(ROOT, "tests", "simt_emu", "_build", "libde265_mi355x_emu.so")


@pytest.fixture(scope="session")
def emu_lib():
    # (pytest-xdist: every worker process has its own session — one build at a time, the others find it up to date)
    import fcntl
    with open(os.path.join(ROOT, "tests", "simt_emu", ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "libde265_amd", "csrc"), "-j8", "emu"], check=True,
                       stdout=subprocess.DEVNULL)
    return capi.Library(EMU_SO)


def run_stream(ctx, pics, n, stages=worklist.STAGE_ALL):
    ctx.set_stages(stages)

    def decode(pic, dst, refs):
        pic.dst_frame = dst
        rf = [-1] * worklist.MAX_REF_FRAMES
        for s, f in refs.items():
            rf[s] = f
        saved = pic.ref_frames
        pic.ref_frames = rf
        ctx.submit(pic)
        ctx.wait()
        pic.ref_frames = saved
        return 0

    # frames are keyed by DPB index in the fixture; map them to context frame handles
    dpb = [p.dst_frame for p in pics]
    for k, (i, pic, planes) in enumerate(replay(decode, ctx.frame_create_for, ctx.frame_destroy, ctx.frame_download, pics[:n])):
        pic.dst_frame = dpb[i]
        assert plane_md5s(planes) == pic.meta["md5"], "picture %d (POC %d) differs from the reference" % (i, pic.meta["poc"])


@pytest.mark.parametrize("variant,n", [("full", 75), ("nolf", 20), ("nosao", 12), ("nodeblk", 12)])
def test_emulated_kernels_match_reference_on_girlshy(emu_lib, variant, n):
    hdr, pics = load_gold("girlshy_%s.m355gold.gz" % variant)
    ctx = capi.Context(emu_lib, 0)
    try:
        run_stream(ctx, pics, n)
    finally:
        ctx.close()
