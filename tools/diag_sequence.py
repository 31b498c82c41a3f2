#!/usr/bin/env python3
"""A soak worker's SEQUENCE again (tools/soak_gpu.py: worker k of n takes seeds first + k, first + k + n, ...; one context for all of them), for differences that only
show in a context with a history.  On the first picture that differs: the same picture again in the same context, by stage mask, and in a fresh context.
python tools/diag_sequence.py <first> <k> <n> <last seed>   (SOAK_SCALE as in the soak; SEQ_ONLY_DEPTH1=1 leaves the three-in-flight decodes out)"""
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


def diff(got, want):
    msg = []
    for c, (g, w) in enumerate(zip(got, want)):
        d = np.argwhere(g != w)
        if len(d):
            msg.append("plane %d: %d differ, y %d..%d x %d..%d, first (%d,%d) got %d want %d" % (c, len(d), d[:, 0].min(), d[:, 0].max(), d[:, 1].min(), d[:, 1].max(), d[0][1], d[0][0], g[tuple(d[0])], w[tuple(d[0])]))
    return "; ".join(msg)


if __name__ == "__main__":
    first, k, n, last = (int(a) for a in sys.argv[1:5])
    lib = capi.Library()
    o = Oracle(ctypes.CDLL(os.path.join(ROOT, "oracle", "liboracle.so")))
    ctx = capi.Context(lib, 0)
    scale = int(os.environ.get("SOAK_SCALE", "1"))
    failures = 0
    for seed in range(first + k, last + 1, n):
        case = random_case(seed)
        case["width"] *= scale; case["height"] *= scale
        try:
            pic, refs = make_case(**case)
        except RuntimeError:
            continue
        want = oracle_decode(o, pic, refs)
        ctx.set_pipeline_depth(1)
        m1 = diff(device_decode(ctx, pic, refs), want)
        m3 = ""
        if not os.environ.get("SEQ_ONLY_DEPTH1"):
            ctx.set_pipeline_depth(3)
            m3 = diff(device_decode(ctx, pic, refs, resident=True, repeat=4), want)
            ctx.set_pipeline_depth(1)
        print("seed %d %dx%d cf %d bd %d ctb %d tiles %dx%d slices %d feat %d deblock %d sao %d: %d pbs %d ibs | depth 1: %s | depth 3: %s" % (
            seed, case["width"], case["height"], case["chroma_format"], case["bit_depth"], 1 << case["log2_ctb"], case["tile_cols"], case["tile_rows"], case["n_slices"], case["features"], case["deblock"],
            case["sao"], len(pic.pbs), len(pic.ibs), m1 or "identical", m3 or "identical"), flush=True)
        if m1 or m3:
            failures += 1
            print("   again, same context, depth 1: %s" % (diff(device_decode(ctx, pic, refs), want) or "identical"))
            for stages in (1, 3, 7, 15, 31):
                w2 = oracle_decode(o, pic, refs, stages)
                print("   stages %2d, same context: %s" % (stages, diff(device_decode(ctx, pic, refs, stages), w2) or "identical"))
            c2 = capi.Context(lib, 0)
            print("   fresh context: %s" % (diff(device_decode(c2, pic, refs), want) or "identical"), flush=True)
            c2.close()
            if failures >= int(os.environ.get("SEQ_MAX_FAILURES", "3")):
                break
