#!/usr/bin/env python3
"""tools/soak_gpu.py's one-picture-at-a-time decode with a second look: when the downloaded frame differs from the oracle's, the SAME device frame is downloaded again
(no decode in between) and — if still different — the picture is decoded once more into it.  Tells a decode that went wrong from a download that ran ahead of the
decode's last stores.  python tools/soak_recheck.py <first seed> <count> [processes]   (SOAK_SCALE as in soak_gpu.py)"""
import ctypes
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def ndiff(got, want):
    return [int((g != w).sum()) for g, w in zip(got, want)]


def work(args):
    first, count, k, n = args
    from libde265_amd import capi, worklist
    from oracle_py import Oracle
    from synth_util import make_case, oracle_decode
    from test_gpu_random import random_case
    lib = capi.Library()
    o = Oracle(ctypes.CDLL(os.path.join(ROOT, "oracle", "liboracle.so")))
    ctx = capi.Context(lib, 0)
    ctx.set_pipeline_depth(1)
    scale = int(os.environ.get("SOAK_SCALE", "1"))
    done = 0
    out = []
    for seed in range(first + k, first + count, n):
        case = random_case(seed)
        case["width"] *= scale; case["height"] *= scale
        try:
            pic, refs = make_case(**case)
        except RuntimeError:
            continue
        want = oracle_decode(o, pic, refs)
        pp = pic.pp[0]
        handles = []
        for planes in refs:
            f = ctx.frame_create_for(pp)
            ctx.frame_upload(f, planes)
            handles.append(f)
        dst = ctx.frame_create_for(pp)
        pic.dst_frame = dst
        pic.ref_frames = [handles[i] if i < len(handles) else -1 for i in range(worklist.MAX_REF_FRAMES)]
        import hashlib
        want_md5 = [hashlib.md5(a.tobytes()).digest() for a in want]

        def look(tag, redo):
            d1 = ndiff(ctx.frame_download(dst), want)
            if not any(d1):
                return
            h1 = [int(a != b) for a, b in zip(ctx.frame_hash(dst, capi.HASH_MD5), want_md5)]      # the device's own MD5 of the frame (k_hash): 0 = the plane is right on the device
            d2 = ndiff(ctx.frame_download(dst), want)
            da = ndiff(ctx.frame_download_finish(ctx.frame_download_async(dst)), want)             # the copies queued on the writer's stream
            time.sleep(0.05)
            d3 = ndiff(ctx.frame_download(dst), want)
            # the reference frames as they are on the device (asynchronous download) against what was uploaded
            dr = [sum(ndiff(ctx.frame_download_finish(ctx.frame_download_async(hf)), planes)) for hf, planes in zip(handles, refs)]
            redo()
            d4 = ndiff(ctx.frame_download(dst), want)
            h4 = [int(a != b) for a, b in zip(ctx.frame_hash(dst, capi.HASH_MD5), want_md5)]
            out.append((seed, tag, "%dx%d cf %d bd %d" % (case["width"], case["height"], case["chroma_format"], case["bit_depth"]), d1, "dev-md5-differs", h1, d2, "async", da, d3, "refs-on-device-differ", dr, "again", d4, h4))

        def once():
            ctx.submit(pic); ctx.wait()
        once()
        look("depth 1", once)
        if not os.environ.get("RECHECK_ONLY_DEPTH1"):
            # (tools/soak_gpu.py's second half: the resident lists four times with three in flight, into the same frame)
            ctx.set_pipeline_depth(3)
            h = ctx.upload(pic)

            def four():
                for _ in range(4):
                    ctx.decode_resident(h)
                ctx.wait()
            four()
            look("depth 3", four)
            ctx.release(h)
            ctx.set_pipeline_depth(1)
        ctx.frame_destroy(dst)
        for f in handles:
            ctx.frame_destroy(f)
        done += 1
    ctx.close()
    return done, out


if __name__ == "__main__":
    first, count = int(sys.argv[1]), int(sys.argv[2])
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    t0 = time.time()
    with mp.get_context("spawn").Pool(n) as pool:
        res = pool.map(work, [(first, count, k, n) for k in range(n)])
    done = sum(r[0] for r in res); bad = [b for r in res for b in r[1]]
    print("soak_recheck: seeds %d..%d, %d pictures, %d differ at the first download, %.0f s on %d processes" % (first, first + count - 1, done, len(bad), time.time() - t0, n))
    print("  (seed, geometry, samples that differ per plane: first download | does the device's own MD5 of each plane differ | second download | asynchronous download | third after 50 ms | after decoding again + MD5)")
    for b in bad[:40]:
        print("  ", b)
