#!/usr/bin/env python3
# NOTE: the M355_X_PROF hooks this tool reads left the product sources in round 6 — apply tools/experiments/product_experiment_hooks_r5.patch to a scratch copy first.
"""Per-CTB timeline of k_intra on an inter picture (experiment build -DM355_X_PROF=100000):
M355_LIB=libde265_amd/variants/prof.so python tools/prof_timeline_sparse.py [workload]
When each CTB with intra blocks was claimed, started its block loop, ended it, was written out (100 MHz wall clock)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from libde265_amd import capi, synth, worklist
lib = capi.Library(); ctx = capi.Context(lib, 0)
w = sys.argv[1] if len(sys.argv) > 1 else "c3_4k_inter"
pic = synth.picture(**synth.CONFIGS[w]); pp = pic.pp[0]
refs = []
for i in range(2):
    f = ctx.frame_create_for(pp); refs.append(f)
pic.ref_frames = (refs + [-1] * worklist.MAX_REF_FRAMES)[:worklist.MAX_REF_FRAMES]
pic.dst_frame = ctx.frame_create_for(pp); h = ctx.upload(pic); ctx.wait()
for _ in range(3): ctx.decode_resident(h)
ctx.wait()
ctx.decode_resident(h); ctx.wait()
N = 65536
buf = (ctypes.c_uint64 * N)()
lib.lib.m355_x_prof_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
rc = lib.lib.m355_x_prof_read(buf, N)
a = np.frombuffer(buf, np.uint64).astype(np.int64)[8200:8200 + 5 * 9000].reshape(-1, 5)
idx = np.nonzero(a[:, 0] > 0)[0]
a = a[idx]
if os.environ.get("TL_OUT"): np.save(os.environ["TL_OUT"], np.concatenate([idx[:, None], a], 1))
T0 = a[:, 1].min()
t = (a[:, 1:] - T0) / 100.0
print("%s: %d CTBs with intra blocks; k_intra span %.1f us" % (w, len(a), t[:, 3].max()))
for name, v in (("claim time", t[:, 0]), ("prologue (claim -> loop)", t[:, 1] - t[:, 0]), ("block loop (incl. waiting)", t[:, 2] - t[:, 1]), ("write-out", t[:, 3] - t[:, 2]), ("whole CTB", t[:, 3] - t[:, 0])):
    print("%-28s median %6.1f  p90 %6.1f  p99 %6.1f  max %6.1f us" % (name, np.median(v), np.percentile(v, 90), np.percentile(v, 99), v.max()))
h_, e = np.histogram(t[:, 0], bins=12)
print("claims over time (us):", " ".join("%d@%.0f" % (c, x) for c, x in zip(h_, e[:-1])))
h_, e = np.histogram(t[:, 3], bins=12)
print("ends over time   (us):", " ".join("%d@%.0f" % (c, x) for c, x in zip(h_, e[:-1])))
order = np.argsort(-(t[:, 2] - t[:, 1]))[:8]
print("longest block loops: ", " ".join("item %d: %.1f us (start %.1f)" % (idx[i], t[i, 2] - t[i, 1], t[i, 1]) for i in order))
