#!/usr/bin/env python3
"""Soak of the product library on the GPU against the CPU oracle (the checker) on random pictures BEYOND the test suite's 200 seeds: geometry, bit depth, chroma format,
CTB size, tiling, slicing, block mix and the optional coding tools drawn at random (tests/test_gpu_random.py random_case), one picture at a time and four decodes of the
resident lists with three in flight.  python tools/soak_gpu.py <first seed> <count> [processes]  ->  one line per process + a total; exit code 1 on any difference.
SOAK_SCALE=<n> multiplies every picture's width and height by n (random_case draws up to 640 x 360): the launch orders the runtime picks by SIZE — the zero fill inside
k_job_count from 16 384 prediction blocks, the two-stream lanes above 16 Mi samples — are then drawn too (n = 8: up to 5120 x 2880; n = 12: up to 7680 x 4320)."""
import ctypes
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def work(args):
    first, count, k, n = args
    from libde265_amd import capi
    from oracle_py import Oracle
    from synth_util import assert_planes_equal, device_decode, make_case, oracle_decode
    from test_gpu_random import random_case
    lib = capi.Library()
    o = Oracle(ctypes.CDLL(os.path.join(ROOT, "oracle", "liboracle.so")))
    ctx = capi.Context(lib, 0)
    done = skipped = 0
    bad = []
    for seed in range(first + k, first + count, n):
        case = random_case(seed)
        scale = int(os.environ.get("SOAK_SCALE", "1"))
        if scale > 1:
            case["width"] *= scale; case["height"] *= scale
        try:
            pic, refs = make_case(**case)
        except RuntimeError:
            skipped += 1
            continue
        want = oracle_decode(o, pic, refs)
        try:
            ctx.set_pipeline_depth(1)
            assert_planes_equal(device_decode(ctx, pic, refs), want, "seed %d depth 1" % seed)
            ctx.set_pipeline_depth(3)
            assert_planes_equal(device_decode(ctx, pic, refs, resident=True, repeat=4), want, "seed %d depth 3" % seed)
        except AssertionError as e:
            bad.append((seed, str(e)[:200]))
        except Exception as e:                              # noqa: BLE001  (an error of the library: reported, the context is made anew)
            bad.append((seed, "exception %s: %s %r" % (type(e).__name__, str(e)[:200], case)))
            try:
                ctx.close()
            except Exception:                               # noqa: BLE001
                pass
            ctx = capi.Context(lib, 0)
        done += 1
    ctx.close()
    return done, skipped, bad


if __name__ == "__main__":
    first, count = int(sys.argv[1]), int(sys.argv[2])
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    t0 = time.time()
    with mp.get_context("spawn").Pool(n) as pool:
        res = pool.map(work, [(first, count, k, n) for k in range(n)])
    done = sum(r[0] for r in res); skipped = sum(r[1] for r in res); bad = [b for r in res for b in r[2]]
    print("soak_gpu: seeds %d..%d, %d pictures decoded twice (one at a time, three in flight), %d generator refusals, %d DIFFER, %.0f s on %d processes"
          % (first, first + count - 1, done, skipped, len(bad), time.time() - t0, n))
    for seed, msg in bad[:20]:
        print("  seed %d: %s" % (seed, msg))
    sys.exit(1 if bad else 0)
