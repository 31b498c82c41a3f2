#!/usr/bin/env python3
"""Soak of the TILE-SHARDED decode on the GPU against the CPU oracle (the checker): the random pictures of tests/test_gpu_random.py random_case with a random tile
grid of at least two tiles forced on them (uniform spacing, 2..4 x 1..3, filtering across tile boundaries on or off as drawn), decoded by an in-process group of
2..min(8, tiles) virtual ranks (m355_group_*: one context per rank, all on the one GPU; the exchanges X0..X3 are peer copies ordered by events) with 1..3 pictures in
flight, every rank's gathered frame compared.  python tools/soak_shard.py <first seed> <count> [processes]  ->  a summary line; exit code 1 on any difference."""
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
    from shard_util import group_sharded_decode
    from synth_util import assert_planes_equal, make_case, oracle_decode
    from test_gpu_random import random_case
    lib = capi.Library()
    o = Oracle(ctypes.CDLL(os.path.join(ROOT, "oracle", "liboracle.so")))
    done = skipped = 0
    bad = []
    for seed in range(first + k, first + count, n):
        case = random_case(seed)
        scale = int(os.environ.get("SOAK_SCALE", "1"))     # (as in tools/soak_gpu.py: every picture's width and height times n)
        case["width"] *= scale; case["height"] *= scale
        rng = np.random.default_rng(31000 + seed)
        ctbs_x, ctbs_y = -(-case["width"] >> case["log2_ctb"]), -(-case["height"] >> case["log2_ctb"])
        tc, tr = int(rng.integers(1, min(4, ctbs_x) + 1)), int(rng.integers(1, min(3, ctbs_y) + 1))
        if tc * tr < 2:
            tc = min(2, ctbs_x); tr = 2 if tc < 2 else tr
        if tc * tr < 2 or tr > ctbs_y:
            skipped += 1
            continue
        case.update(tile_cols=tc, tile_rows=tr)
        nranks = int(rng.integers(2, min(8, tc * tr) + 1))
        depth = int(rng.integers(1, 4))
        try:
            pic, refs = make_case(**case)
        except RuntimeError:
            skipped += 1
            continue
        want = oracle_decode(o, pic, refs)
        try:
            for r, got in enumerate(group_sharded_decode(lib, pic, refs, nranks, depth=depth, repeat=2)):
                assert_planes_equal(got, want, "seed %d: %dx%d tiles, rank %d of %d, depth %d" % (seed, tc, tr, r, nranks, depth))
        except AssertionError as e:
            bad.append((seed, str(e)[:240]))
        except Exception as e:                              # noqa: BLE001
            bad.append((seed, "exception %s: %s (%dx%d tiles, %d ranks, depth %d, %r)" % (type(e).__name__, str(e)[:160], tc, tr, nranks, depth, case)))
        done += 1
    return done, skipped, bad


if __name__ == "__main__":
    first, count = int(sys.argv[1]), int(sys.argv[2])
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    t0 = time.time()
    with mp.get_context("spawn").Pool(n) as pool:
        res = pool.map(work, [(first, count, k, n) for k in range(n)])
    done = sum(r[0] for r in res); skipped = sum(r[1] for r in res); bad = [b for r in res for b in r[2]]
    print("soak_shard: seeds %d..%d, %d pictures decoded by 2..8 virtual ranks (twice each, 1..3 in flight), %d not built, %d DIFFER, %.0f s on %d processes"
          % (first, first + count - 1, done, skipped, len(bad), time.time() - t0, n))
    for seed, msg in bad[:20]:
        print("  seed %d: %s" % (seed, msg))
    sys.exit(1 if bad else 0)
