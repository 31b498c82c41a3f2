#!/usr/bin/env python3
"""ONE context for a long time (what a decoder that runs for hours does to the library's rings and counters): <n> rounds of
  * 2000 decodes of resident lists with three in flight, two pictures' handles and four destination frames going round (event ring, status ring, lanes' scratch),
  * 200 submits of the same lists through the copying entry and 200 through the in-place arena (transient arenas, validation),
  * 50 frame create / upload / destroy cycles and 50 list upload / release cycles (handle reuse),
with the destination frames checked against the oracle every round, the process's resident set and the device's free memory (rocm-smi is not asked: hipMemGetInfo
through ctypes on libamdhip64) printed at the start and at the end.  python tools/soak_long.py <rounds> [picture size multiplier]"""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from libde265_amd import capi, worklist                           # noqa: E402
from oracle_py import Oracle                                      # noqa: E402
from synth_util import assert_planes_equal, make_case, oracle_decode   # noqa: E402


def rss_mb():
    for ln in open("/proc/self/status"):
        if ln.startswith("VmRSS"):
            return int(ln.split()[1]) / 1024.0
    return 0.0


def dev_free_mb():
    try:
        hip = ctypes.CDLL("libamdhip64.so")
        f, t = ctypes.c_size_t(0), ctypes.c_size_t(0)
        if hip.hipMemGetInfo(ctypes.byref(f), ctypes.byref(t)) == 0:
            return f.value / 2 ** 20
    except OSError:
        pass
    return -1.0


if __name__ == "__main__":
    rounds = int(sys.argv[1])
    mul = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    lib = capi.Library()
    o = Oracle(ctypes.CDLL(os.path.join(ROOT, "oracle", "liboracle.so")))
    ctx = capi.Context(lib, 0)
    ctx.set_pipeline_depth(3)
    cases = [dict(width=416 * mul, height=240 * mul, bit_depth=8, seed=5001, n_refs=2, tile_cols=2), dict(width=416 * mul, height=240 * mul, bit_depth=8, seed=5002, n_refs=2, intra_pct=30, n_slices=2)]
    pics, want = [], []
    refs = None
    for c in cases:
        p, r = make_case(**c)
        refs = refs or r                                        # (both pictures predicted from the first one's references: same geometry)
        pics.append(p)
        want.append(oracle_decode(o, p, refs))
    pp = pics[0].pp[0]
    rf = []
    for planes in refs:
        f = ctx.frame_create_for(pp); ctx.frame_upload(f, planes); rf.append(f)
    dsts = [ctx.frame_create_for(pp) for _ in range(4)]
    handles = {}
    for i, p in enumerate(pics):
        for d in dsts:
            p.dst_frame = d
            p.ref_frames = [rf[k] if k < len(rf) else -1 for k in range(worklist.MAX_REF_FRAMES)]
            handles[(i, d)] = ctx.upload(p)
    ctx.wait()
    t0 = time.time()
    print("start: rss %.0f MB, device free %.0f MB" % (rss_mb(), dev_free_mb()), flush=True)
    n_dec = 0
    N = int(os.environ.get("SOAK_LONG_N", "2000"))          # (a multiple of 4; the CPU tier's interpreter wants a small one)
    for rnd in range(rounds):
        for k in range(N):
            ctx.decode_resident(handles[(k & 1, dsts[k % 4])]); n_dec += 1
        ctx.wait()
        for j, d in enumerate(dsts):                             # frame d was last written by decode k = N - 4 + j: picture j & 1
            assert_planes_equal(ctx.frame_download(d), want[j & 1], "round %d frame %d" % (rnd, j))
        for k in range(N // 10):
            p = pics[k & 1]; p.dst_frame = dsts[k % 4]
            ctx.submit(p); n_dec += 1
        ctx.wait()
        for k in range(N // 10):
            p = pics[k & 1]; p.dst_frame = dsts[k % 4]
            h = ctx.upload_in_place(p, slack=1.2)
            ctx.decode_resident(h); n_dec += 1
            ctx.release(h)
        ctx.wait()
        for j, d in enumerate(dsts):
            assert_planes_equal(ctx.frame_download(d), want[(N // 10 - 4 + j) & 1], "round %d frame %d after the submits" % (rnd, j))
        for k in range(max(2, N // 40)):
            f = ctx.frame_create_for(pp); ctx.frame_upload(f, refs[0]); ctx.frame_destroy(f)
            p = pics[k & 1]; p.dst_frame = dsts[0]
            h = ctx.upload(p); ctx.release(h)
        if rnd % 10 == 9 or rnd == rounds - 1:
            print("round %d: %d decodes, rss %.0f MB, device free %.0f MB, %.0f s" % (rnd + 1, n_dec, rss_mb(), dev_free_mb(), time.time() - t0), flush=True)
    ctx.close()
    print("soak_long: %d decodes in one context, every check identical" % n_dec)
