#!/usr/bin/env python3
# NOTE: the M355_X_PROF hooks this tool reads left the product sources in round 6 — apply tools/experiments/product_experiment_hooks_r5.patch to a scratch copy first.
"""Per-CTB timeline of k_intra on an all-intra picture (experiment build -DM355_X_PROF=100000):
M355_LIB=libde265_amd/variants/prof.so python tools/prof_timeline.py [cu_log2]
For every CTB: when its workgroup claimed it, when its block loop started (prologue done), when its last level ended, when its
samples were written out (100 MHz wall clock).  Prints the wavefront step statistics: how long after its left / top-right
neighbour's last level a CTB's last level ends, and how much of that it spent in its own levels."""
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
a = np.frombuffer(buf, np.uint64).astype(np.int64)[8200:8200 + 5 * 1600].reshape(-1, 5)
a = a[a[:, 0] > 0]
if os.environ.get('TL_OUT'): np.save(os.environ['TL_OUT'], a)
W = (int(pp["width"]) + 63) // 64; H = (int(pp["height"]) + 63) // 64
t = {}
for ctb1, t0, t1, t2, t3 in a: t[int(ctb1) - 1] = (t0, t1, t2, t3)
T0 = min(v[0] for v in t.values()); Tend = max(v[3] for v in t.values())
print("cu_log2=%d: %d CTBs (%dx%d), k_intra span %.1f us" % (cu, len(t), W, H, (Tend - T0) / 100.0))
pro = np.array([v[1] - v[0] for v in t.values()]) / 100.0; own = np.array([v[2] - v[1] for v in t.values()]) / 100.0; wr = np.array([v[3] - v[2] for v in t.values()]) / 100.0
print("prologue (claim -> loop)   median %.1f us  p90 %.1f  max %.1f" % (np.median(pro), np.percentile(pro, 90), pro.max()))
print("block loop (incl. waiting) median %.1f us  p90 %.1f  max %.1f" % (np.median(own), np.percentile(own, 90), own.max()))
print("write-out                  median %.1f us  p90 %.1f  max %.1f" % (np.median(wr), np.percentile(wr, 90), wr.max()))
steps = []; late = []
for c, v in t.items():
    x, y = c % W, c // W
    deps = []
    if x > 0 and (c - 1) in t: deps.append(t[c - 1][2])
    if y > 0 and x + 1 < W and (c - W + 1) in t: deps.append(t[c - W + 1][2])
    elif y > 0 and (c - W) in t: deps.append(t[c - W][2])
    if not deps: continue
    steps.append((v[2] - max(deps)) / 100.0)
    late.append((v[1] - max(deps)) / 100.0)     # > 0: the loop started after the neighbours were done (prologue on the critical path)
steps = np.array(steps); late = np.array(late)
print("last level ends this long after the later of (left, top-right) neighbour's: median %.2f us  mean %.2f  p10 %.2f  p90 %.2f" % (np.median(steps), steps.mean(), np.percentile(steps, 10), np.percentile(steps, 90)))
print("CTBs whose loop started after their neighbours had finished: %d of %d (median lateness of those %.1f us)" % ((late > 0).sum(), len(late), np.median(late[late > 0]) if (late > 0).any() else 0))
print("critical path estimate: (W + 2(H-1)) = %d steps x median step = %.1f us" % (W + 2 * (H - 1), (W + 2 * (H - 1)) * np.median(steps)))
# the first row: pure left-neighbour chain
r0 = [(t[x][2] - t[x - 1][2]) / 100.0 for x in range(1, W) if x in t and x - 1 in t]
print("row 0 left-neighbour steps (us):", " ".join("%.1f" % v for v in r0))
