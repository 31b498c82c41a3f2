#!/usr/bin/env python3
"""Which STAGE of a random picture (tools/soak_gpu.py seed, SOAK_SCALE) comes out different from the oracle, where, and is it the same on every run?  For each seed the
stage masks 1 (prediction), 3 (+ residuals), 7 (+ intra), 15 (+ deblocking), 31 (+ SAO) are decoded by the oracle and <reps> times by the library, one picture at a time.
python tools/diag_picture.py <reps> <seed> [seed ...]   (environment: SOAK_SCALE, and whatever the library reads, e.g. M355_CLEAR_IN_COUNT_MIN)"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from libde265_amd import capi                                    # noqa: E402
from oracle_py import Oracle                                     # noqa: E402
from synth_util import device_decode, make_case, oracle_decode   # noqa: E402
from test_gpu_random import random_case                          # noqa: E402

if __name__ == "__main__":
    reps = int(sys.argv[1])
    lib = capi.Library()
    o = Oracle(ctypes.CDLL(os.path.join(ROOT, "oracle", "liboracle.so")))
    ctx = capi.Context(lib, 0)
    ctx.set_pipeline_depth(int(os.environ.get("DIAG_DEPTH", "1")))
    scale = int(os.environ.get("SOAK_SCALE", "1"))
    for seed in map(int, sys.argv[2:]):
        case = random_case(seed)
        case["width"] *= scale; case["height"] *= scale
        pic, refs = make_case(**case)
        print("seed %d %r: %d cus %d pbs %d tus %d ibs rbs %s" % (seed, case, len(pic.cus), len(pic.pbs), len(pic.tus), len(pic.ibs), list(pic.rb_count)))
        for stages in (1, 3, 7, 15, 31):
            want = oracle_decode(o, pic, refs, stages)
            for rep in range(reps):
                got = device_decode(ctx, pic, refs, stages)
                msg = []
                for c, (g, w) in enumerate(zip(got, want)):
                    d = np.argwhere(g != w)
                    if len(d):
                        msg.append("plane %d: %d differ, y %d..%d x %d..%d, first (%d,%d) got %d want %d" % (c, len(d), d[:, 0].min(), d[:, 0].max(), d[:, 1].min(), d[:, 1].max(), d[0][1], d[0][0], g[tuple(d[0])], w[tuple(d[0])]))
                print("   stages %2d run %d: %s" % (stages, rep, "; ".join(msg) or "identical"))
