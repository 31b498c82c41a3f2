#!/usr/bin/env python3
# NOTE: the M355_X_PROF hooks this tool reads left the product sources in round 6 — apply tools/experiments/product_experiment_hooks_r5.patch to a scratch copy first.
"""Per-workgroup timeline of k_inter_jobs (experiment build -DM355_X_PROF=100000):
M355_LIB=libde265_amd/variants/prof.so python tools/prof_inter_timeline.py [workload]
For every workgroup: when it entered, when it knew its class, when its tables were in LDS, when its luma was written, when it was done
(100 MHz wall clock).  Prints the launch's span and, per job class, how long the phases of a workgroup take and when workgroups start."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from libde265_amd import capi, synth, worklist
lib = capi.Library(); ctx = capi.Context(lib, 0)
name = sys.argv[1] if len(sys.argv) > 1 else "c3_4k_inter"
cfg = dict(synth.CONFIGS[name]); pic = synth.picture(**cfg); pp = pic.pp[0]
refs = []
for i in range(cfg["n_refs"]):
    f = ctx.frame_create_for(pp); ctx.frame_upload(f, synth.ref_planes(cfg["seed"] + 17 * i, int(pp["width"]), int(pp["height"]), 1, int(pp["bit_depth_luma"]))); refs.append(f)
pic.ref_frames = [refs[i] if i < len(refs) else -1 for i in range(worklist.MAX_REF_FRAMES)]
pic.dst_frame = ctx.frame_create_for(pp); h = ctx.upload(pic); ctx.wait()
for _ in range(3): ctx.decode_resident(h)
ctx.wait()
ctx.decode_resident(h); ctx.wait()
buf = (ctypes.c_uint64 * 131072)()
lib.lib.m355_x_prof_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
lib.lib.m355_x_prof_read(buf, 131072)
a = np.frombuffer(buf, np.uint64).astype(np.int64)[65536:65536 + 6 * 8100].reshape(-1, 6)
ent = a[a[:, 0] > 0]
T0 = ent[:, 0].min()
work = a[a[:, 5] > 0]
print("%s: %d workgroups entered, %d with a class; entries over %.1f us" % (name, len(ent), len(work), (ent[:, 0].max() - T0) / 100.0))
done = work[work[:, 4] > 0]
print("launch span (first entry -> last job done): %.1f us" % ((done[:, 4].max() - T0) / 100.0))
for cls, nm in ((1, "uni"), (2, "bi"), (3, "weighted"), (4, "edge")):
    w = done[done[:, 5] == cls]
    if not len(w): continue
    st = (w[:, 0] - T0) / 100.0
    ph = [(w[:, 1] - w[:, 0]) / 100.0, (w[:, 2] - w[:, 1]) / 100.0, (w[:, 3] - w[:, 2]) / 100.0, (w[:, 4] - w[:, 3]) / 100.0, (w[:, 4] - w[:, 0]) / 100.0]
    print("%-8s %5d workgroups  start median %.1f us (p90 %.1f max %.1f)  end max %.1f" % (nm, len(w), np.median(st), np.percentile(st, 90), st.max(), ((w[:, 4] - T0) / 100.0).max()))
    for lab, v in zip(("entry->class", "class->tables", "tables->luma", "luma->done", "whole"), ph):
        print("         %-14s median %6.2f us  p10 %6.2f  p90 %6.2f  max %6.2f" % (lab, np.median(v), np.percentile(v, 10), np.percentile(v, 90), v.max()))
# concurrency: how many workgroups with a class are between entry and done at a time
ev = sorted([(t, 1) for t in done[:, 0]] + [(t, -1) for t in done[:, 4]])
cur = peak = 0; area = 0; last = ev[0][0]
for t, d in ev:
    area += cur * (t - last); last = t; cur += d; peak = max(peak, cur)
print("working workgroups in flight: peak %d, mean %.0f over the span" % (peak, area / max(1, done[:, 4].max() - done[:, 0].min())))
