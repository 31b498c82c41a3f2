#!/usr/bin/env python3
# NOTE: the M355_X_PROF hooks this tool reads left the product sources in round 6 — apply tools/experiments/product_experiment_hooks_r5.patch to a scratch copy first.
"""Per-level timing of one CTB of k_intra (experiment build -DM355_X_PROF=<work item>): M355_LIB=libde265_amd/variants/prof.so python tools/prof_intra.py [cu_log2]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from libde265_amd import capi, synth, worklist
lib = capi.Library(); ctx = capi.Context(lib, 0)
cu = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cfg = dict(synth.CONFIGS["c2_1080p_intra"], fixed_cu_log2=cu)
pic = synth.picture(**cfg); pp = pic.pp[0]
pic.ref_frames = [-1] * worklist.MAX_REF_FRAMES
pic.dst_frame = ctx.frame_create_for(pp); h = ctx.upload(pic); ctx.wait()
for _ in range(3): ctx.decode_resident(h)
ctx.wait()
ctx.decode_resident(h); ctx.wait()
buf = (ctypes.c_uint64 * 16384)()
lib.lib.m355_x_prof_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
rc = lib.lib.m355_x_prof_read(buf, 16384)
a = np.frombuffer(buf, np.uint64).astype(np.int64)
n = int(a[3])
print("rc", rc, "levels recorded", n)
rows = a[4:4 + 8 * min(n, 1023)].reshape(-1, 8)
t0c, t0w, t1w = a[0], a[1], a[2]
tend = rows[-1, 4]
print("cycle counter ticks %d, wall ticks (100 MHz) %d -> counter runs at %.1f MHz; whole loop %.1f us" % (tend - t0c, t1w - t0w, (tend - t0c) / max(1, (t1w - t0w)) * 100.0, (t1w - t0w) / 100.0))
d = np.stack([rows[:, 1] - rows[:, 0], rows[:, 2] - rows[:, 1], rows[:, 3] - rows[:, 2], rows[:, 4] - rows[:, 3], rows[:, 4] - rows[:, 0]], 1)
has = rows[:, 1] > 0
print("levels where wave 0 had a block: %d of %d" % (has.sum(), len(rows)))
for name, col in (("gather", 0), ("predict+store", 1), ("loop tail", 2), ("barrier wait", 3), ("level total", 4)):
    x = d[has, col] if col < 3 else d[:, col]
    print("%-14s median %6d  mean %8.1f  p90 %6d  max %7d ticks" % (name, np.median(x), x.mean(), np.percentile(x, 90), x.max()))
print("first 40 levels (gather, predict, tail, barrier, total):")
for r_, dd in zip(rows[:40], d[:40]): print("  L%-3d %s%s" % (int(r_[5] & 0xFFFFFFFF), " ".join("%6d" % v for v in dd), "" if r_[1] > 0 else "   (no block)"))
