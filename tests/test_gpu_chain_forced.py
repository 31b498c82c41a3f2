"""GPU parity of the schedules the runtime picks BY TIMING (runtime_decode.hip decode_pre / launch_prediction): a picture whose newest reference is
still being written takes the chain schedule — front part on a spare lane, change of stream in front of k_inter, destination hazards behind it,
(up to 4K) residual tiles in the front part + k_residual_add — and which pictures do depends on how the race between the host and the GPU falls.
These tests take the race out:

  * M355_TEST_CHAIN_LANES=1 / M355_TEST_CHAIN_RESIDUALS=1 (test hooks of the runtime, read once per process) in a process of their own: the
    picture suites — every synthetic feature case, the BASELINE configurations at full size, 50 random pictures, the recorded girlshy stream —
    with three pictures in flight, bit for bit against the oracle;
  * the TWO-STREAM-LANE chain (pictures above 16 Mpx fork a side stream, launch_prediction `single`): four different pictures, each predicted from
    the two decoded before it, three frames going round (a picture overwrites the frame the picture before it still reads), depths 2 / 3 / 5, with
    and without SAO, at 4096x4112 8-bit and at the 8K 10-bit 8-tile geometry — as the race falls AND with the hook.

The order the reference guarantees: a picture is complete before the next one reads it (decctx.cc:577-650, motion.cc:288-730)."""
import os
import subprocess
import sys

import pytest

from oracle_py import Oracle
from synth_util import assert_planes_equal
from libde265_amd import capi, synth, worklist

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def chain2_pictures(cfg, n_pics):
    """n_pics different pictures of one geometry (lists only) + the two start frames' planes"""
    pics = [synth.picture(**dict(cfg, seed=cfg["seed"] + 131 * k)) for k in range(n_pics)]
    pp = pics[0].pp[0]
    start = [synth.ref_planes(cfg["seed"] + 17 * i, int(pp["width"]), int(pp["height"]), int(pp["chroma_format_idc"]), int(pp["bit_depth_luma"])) for i in range(2)]
    return pics, start


def chain2_oracle(o, pics, start, n_decodes):
    """decode k: lists k % len(pics), destination frame k % 3, references = the two decodes before it (k = 0, 1: the start frames) -> planes of the three frames at the end"""
    pp = pics[0].pp[0]
    fr = [o.frame_new(pp) for _ in range(3)]
    # frames 1 and 2 hold the start pictures ("decodes -2 and -1"): decode 0 writes frame 0 from (frame 2, frame 1), decode 1 frame 1 from (0, 2), ...
    o.frame_set_planes(fr[1], start[0]); o.frame_set_planes(fr[2], start[1])
    for k in range(n_decodes):
        pic = pics[k % len(pics)]
        pic.ref_frames = [0, 1] + [-1] * (worklist.MAX_REF_FRAMES - 2)
        assert o.decode(pic, fr[k % 3], {0: fr[(k + 2) % 3], 1: fr[(k + 1) % 3]}) == 0
    want = [o.frame_planes(f) for f in fr]
    for f in fr:
        o.frame_free(f)
    return want


def chain2_device(lib, pics, start, n_decodes, depth):
    ctx = capi.Context(lib, 0)
    try:
        ctx.set_pipeline_depth(depth)
        pp = pics[0].pp[0]
        fr = [ctx.frame_create_for(pp) for _ in range(3)]
        ctx.frame_upload(fr[1], start[0]); ctx.frame_upload(fr[2], start[1])
        # one resident copy of the lists per (picture, frame assignment) that occurs: the assignment of decode k depends on k % 3 only
        handles = {}
        for k in range(n_decodes):
            key = (k % len(pics), k % 3)
            if key in handles:
                continue
            pic = pics[key[0]]
            pic.dst_frame = fr[k % 3]
            pic.ref_frames = [fr[(k + 2) % 3], fr[(k + 1) % 3]] + [-1] * (worklist.MAX_REF_FRAMES - 2)
            handles[key] = ctx.upload(pic)
        ctx.wait()
        for k in range(n_decodes):                    # no host synchronisation anywhere in this loop
            ctx.decode_resident(handles[(k % len(pics), k % 3)])
        ctx.wait()
        return [ctx.frame_download(f) for f in fr]
    finally:
        ctx.close()


BIG = {
    "4096x4112_8bit": dict(width=4096, height=4112, bit_depth=8, seed=4401, n_refs=2),                      # just above the one-stream limit (16 Mi samples)
    "8k10_8tiles": dict(width=7680, height=4320, bit_depth=10, seed=4402, n_refs=2, tile_cols=4, tile_rows=2),   # BASELINE's C5 geometry
}
_want = {}


def big_chain_want(oracle, name, sao, n_pics, n_decodes):
    key = (name, sao, n_pics, n_decodes)
    if key not in _want:
        pics, start = chain2_pictures(dict(BIG[name], sao=sao), n_pics)
        _want[key] = (pics, start, chain2_oracle(Oracle(oracle), pics, start, n_decodes))
    return _want[key]


@pytest.mark.parametrize("sao", [1, 0])
@pytest.mark.parametrize("name,n_pics,n_decodes", [("4096x4112_8bit", 4, 8), ("8k10_8tiles", 2, 5)])
def test_two_stream_lane_chain_as_the_race_falls(oracle, name, n_pics, n_decodes, sao):
    pics, start, want = big_chain_want(oracle, name, sao, n_pics, n_decodes)
    lib = capi.Library()
    for depth in (2, 3, 5):
        got = chain2_device(lib, pics, start, n_decodes, depth)
        for i in range(3):
            assert_planes_equal(got[i], want[i], "%s sao %d depth %d: frame %d after %d chained decodes" % (name, sao, depth, i, n_decodes))


WORKER = r"""
import ctypes, os, sys
sys.path.insert(0, %(here)r); sys.path.insert(0, %(root)r)
import test_gpu_chain_forced as T
from oracle_py import Oracle
from synth_util import assert_planes_equal, make_case, oracle_decode
from libde265_amd import capi, synth
o = ctypes.CDLL(%(oracle)r)
lib = capi.Library()
what = sys.argv[1]
if what == "bigchain":
    for name, n_pics, n_decodes in (("4096x4112_8bit", 4, 8), ("8k10_8tiles", 2, 5)):
        for sao in (1, 0):
            pics, start, want = T.big_chain_want(o, name, sao, n_pics, n_decodes)
            for depth in (2, 3, 5):
                got = T.chain2_device(lib, pics, start, n_decodes, depth)
                for i in range(3):
                    assert_planes_equal(got[i], want[i], "%%s sao %%d depth %%d frame %%d" %% (name, sao, depth, i))
    print("forced ok")
elif what == "suites":
    # every picture of the synthetic suites with THREE pictures in flight (the hooks act on lanes: depth >= 2 / >= 3)
    from test_gpu_synth import SMALL
    from test_gpu_random import random_case
    from synth_util import device_decode
    ctx = capi.Context(lib, 0)
    cases = [("small %%d" %% i, c) for i, c in enumerate(SMALL)] + [(n, dict(synth.CONFIGS[n])) for n in ("c2_1080p_intra", "c3_4k_inter", "c4_4k_4tiles", "c5_8k10_8tiles")]
    cases += [("random %%d" %% s, random_case(s)) for s in range(50)]
    n = 0
    for name, case in cases:
        try:
            pic, refs = make_case(**case)
        except RuntimeError:
            continue
        want = oracle_decode(Oracle(o), pic, refs)
        for depth in (3, 2):
            ctx.set_pipeline_depth(depth)
            assert_planes_equal(device_decode(ctx, pic, refs, resident=True, repeat=4), want, "%%s depth %%d" %% (name, depth))
        n += 1
    ctx.close()
    print("forced ok: %%d pictures" %% n)
"""


def run_forced(oracle, what, env_extra, timeout=1500):
    code = WORKER % dict(here=HERE, root=ROOT, oracle=oracle._name)
    r = subprocess.run([sys.executable, "-c", code, what], env=dict(os.environ, **env_extra), capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    assert r.returncode == 0 and "forced ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.parametrize("hooks", [dict(M355_TEST_CHAIN_LANES="1"), dict(M355_TEST_CHAIN_RESIDUALS="1"), dict(M355_TEST_CHAIN_LANES="1", M355_TEST_CHAIN_RESIDUALS="1")],
                         ids=["chain_lanes", "chain_residuals", "both"])
def test_picture_suites_forced_into_the_chain_schedule(oracle, hooks):
    run_forced(oracle, "suites", hooks)


def test_girlshy_and_chains_forced_into_the_chain_schedule(oracle):
    """the recorded stream (real reference chains) and the pipeline tests' chains, every picture in the chain schedule"""
    env = dict(os.environ, M355_TEST_CHAIN_LANES="1", M355_TEST_CHAIN_RESIDUALS="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-p", "no:cacheprovider", "-m", "gpu", os.path.join(HERE, "test_gpu_girlshy.py"),
                        os.path.join(HERE, "test_gpu_pipeline.py"), "-k", "pipelined or destination_the_previous"], env=env, capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert r.returncode == 0 and " passed" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


def test_two_stream_lane_chain_forced(oracle):
    run_forced(oracle, "bigchain", dict(M355_TEST_CHAIN_LANES="1"))
