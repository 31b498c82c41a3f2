#!/usr/bin/env python3
"""Soak of m355_decode_batch (several independent INTRA pictures, one k_intra launch: the pictures' CTB wavefronts interleaved through one ticket) on the GPU against the
CPU oracle: the random geometries / formats / coding tools of tests/test_gpu_random.py random_case as all-intra pictures, 2..12 pictures per context, some of them of
another size (ragged work lists), three rounds of random batches with nothing waited for in between (frames recycled from batch to batch), every picture's last decode
compared (tests/batch_util.py check_batches).  python tools/soak_batch.py <first seed> <count> [processes]   (SOAK_SCALE as in soak_gpu.py)"""
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
    from batch_util import check_batches
    from libde265_amd import capi
    from oracle_py import Oracle
    from test_gpu_random import random_case
    lib = capi.Library()
    o = Oracle(ctypes.CDLL(os.path.join(ROOT, "oracle", "liboracle.so")))
    scale = int(os.environ.get("SOAK_SCALE", "1"))
    done = skipped = 0
    bad = []
    for seed in range(first + k, first + count, n):
        case = random_case(seed)
        rng = np.random.default_rng(91000 + seed)
        case["width"] *= scale; case["height"] *= scale
        case.update(intra_pct=100, n_refs=0, features=case["features"] & ~256)   # (without SYN_MISSING_REF: nothing is predicted from a reference)
        npic = int(rng.integers(2, 13))
        depth = int(rng.integers(npic, 17))
        batches = []
        for _ in range(3):
            order = list(rng.permutation(npic))
            while order:
                m = int(rng.integers(1, len(order) + 1))
                batches.append([int(x) for x in order[:m]]); order = order[m:]
        sizes = {}
        for j in range(npic):
            if rng.random() < 0.25:
                sizes[j] = (int(rng.integers(2, 41)) * 8 * scale, int(rng.integers(2, 24)) * 8 * scale)
        ctx = None
        try:
            ctx = check_batches(lib, o, case, depth, batches, sizes or None)[0]
        except (RuntimeError, ValueError):
            skipped += 1                                    # (a combination the generator does not build)
            continue
        except AssertionError as e:
            bad.append((seed, str(e)[:200] + " %d pictures, %d lanes, batches %r" % (npic, depth, batches)))
        except Exception as e:                              # noqa: BLE001
            bad.append((seed, "exception %s: %s" % (type(e).__name__, str(e)[:200])))
        finally:
            if ctx is not None:
                ctx.close()
        done += 1
    return done, skipped, bad


if __name__ == "__main__":
    first, count = int(sys.argv[1]), int(sys.argv[2])
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    t0 = time.time()
    with mp.get_context("spawn").Pool(n) as pool:
        res = pool.map(work, [(first, count, k, n) for k in range(n)])
    done = sum(r[0] for r in res); skipped = sum(r[1] for r in res); bad = [b for r in res for b in r[2]]
    print("soak_batch: seeds %d..%d, %d contexts of 2..12 intra pictures decoded in random batches (three rounds), %d not built, %d DIFFER, %.0f s on %d processes"
          % (first, first + count - 1, done, skipped, len(bad), time.time() - t0, n))
    for seed, msg in bad[:20]:
        print("  seed %d: %s" % (seed, msg))
    sys.exit(1 if bad else 0)
