#!/usr/bin/env python3
"""Which prediction blocks does k_inter get wrong?  Inter stage only, device vs oracle, on the first synthetic cases: per differing plane the PBs that cover the
differing samples (geometry, vectors, flags) — the view that finds a window / alignment / hazard bug.  usage: python tools/diag_inter.py [case index ...]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle_py import Oracle
from synth_util import device_decode, make_case, oracle_decode
from test_emu_synth import CASES
from libde265_amd import capi, worklist

o = Oracle(ctypes.CDLL(os.path.join(ROOT, "oracle", "liboracle.so")))
lib = capi.Library(os.environ.get("M355_LIB"))
ctx = capi.Context(lib, 0)
which = [int(a) for a in sys.argv[1:]] or list(range(len(CASES)))
for ci in which:
    case = CASES[ci]
    pic, refs = make_case(**case)
    if not len(pic.pbs):
        continue
    st = worklist.STAGE_INTER
    for rep in range(3):
        got, want = device_decode(ctx, pic, refs, st), oracle_decode(o, pic, refs, st)
        bad = [(c, np.argwhere(g != w)) for c, (g, w) in enumerate(zip(got, want)) if not np.array_equal(g, w)]
        print("case %d %r rep %d: %s" % (ci, case, rep, "ok" if not bad else "DIFFERS"))
        for c, d in bad:
            sc = 1 if c == 0 else 2
            seen = set()
            for (y, x) in d[:400]:
                ly, lx = int(y) * sc, int(x) * sc
                for i, pb in enumerate(pic.pbs):
                    if pb["x"] <= lx < pb["x"] + pb["w"] and pb["y"] <= ly < pb["y"] + pb["h"]:
                        seen.add(i)
                        break
            print("  plane %d: %d samples differ, bbox %s..%s, in PBs:" % (c, len(d), d.min(0), d.max(0)))
            for i in sorted(seen)[:12]:
                pb = pic.pbs[i]
                print("    pb %d: x %d y %d %dx%d flags %#x mv0 (%d,%d) mv1 (%d,%d) ref %s  | index in list %d of %d" %
                      (i, pb["x"], pb["y"], pb["w"], pb["h"], pb["flags"], pb["mv"][0][0], pb["mv"][0][1], pb["mv"][1][0], pb["mv"][1][1], list(pb["ref_slot"]), i, len(pic.pbs)))
ctx.close()
