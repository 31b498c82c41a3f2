#!/usr/bin/env python3
"""Soak of DEPENDENT CHAINS on the GPU against the CPU oracle (the checker): the random geometries / coding tools of tests/test_gpu_random.py random_case (two references,
at least some prediction blocks), three different pictures' lists going round, every decode predicted from the two decodes before it, three destination frames going
round (a picture overwrites the frame the picture before it still reads), 5..9 decodes without any host synchronisation, 2..5 lanes (tests/test_gpu_chain_forced.py
chain2_*): the schedules the runtime picks by timing — chain lanes, front-part residuals, late destination hazards — at random sizes and formats.
python tools/soak_chain.py <first seed> <count> [processes]   (SOAK_SCALE as in soak_gpu.py; M355_TEST_CHAIN_LANES / M355_TEST_CHAIN_RESIDUALS force the schedules)"""
import ctypes
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def work(args):
    first, count, k, n = args
    from libde265_amd import capi
    from oracle_py import Oracle
    from synth_util import assert_planes_equal
    from test_gpu_chain_forced import chain2_device, chain2_oracle, chain2_pictures
    from test_gpu_random import random_case
    lib = capi.Library()
    o = Oracle(ctypes.CDLL(os.path.join(ROOT, "oracle", "liboracle.so")))
    scale = int(os.environ.get("SOAK_SCALE", "1"))
    done = skipped = 0
    bad = []
    for seed in range(first + k, first + count, n):
        case = random_case(seed)
        rng = np.random.default_rng(55000 + seed)
        case["width"] *= scale; case["height"] *= scale
        case["n_refs"] = 2
        if case["intra_pct"] == 100:
            case["intra_pct"] = 30
        n_decodes, depth = int(rng.integers(5, 10)), int(rng.integers(2, 6))
        try:
            pics, start = chain2_pictures(case, 3)
        except RuntimeError:
            skipped += 1
            continue
        want = chain2_oracle(o, pics, start, n_decodes)
        try:
            got = chain2_device(lib, pics, start, n_decodes, depth)
            for f in range(3):
                assert_planes_equal(got[f], want[f], "seed %d frame %d (%d decodes, %d lanes)" % (seed, f, n_decodes, depth))
        except AssertionError as e:
            bad.append((seed, str(e)[:220]))
        except Exception as e:                              # noqa: BLE001
            bad.append((seed, "exception %s: %s %r" % (type(e).__name__, str(e)[:160], case)))
        done += 1
    return done, skipped, bad


if __name__ == "__main__":
    first, count = int(sys.argv[1]), int(sys.argv[2])
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    t0 = time.time()
    with mp.get_context("spawn").Pool(n) as pool:
        res = pool.map(work, [(first, count, k, n) for k in range(n)])
    done = sum(r[0] for r in res); skipped = sum(r[1] for r in res); bad = [b for r in res for b in r[2]]
    print("soak_chain: seeds %d..%d, %d chains of 5..9 dependent decodes on 2..5 lanes, %d not built, %d DIFFER, %.0f s on %d processes" % (first, first + count - 1, done, skipped, len(bad), time.time() - t0, n))
    for seed, msg in bad[:20]:
        print("  seed %d: %s" % (seed, msg))
    sys.exit(1 if bad else 0)
