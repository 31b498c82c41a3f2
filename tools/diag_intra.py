#!/usr/bin/env python3
"""Intra stage time vs block structure (1080p all-intra): one picture at a time and with pictures in flight, for the CU sizes that
give different dependency chains.  usage: [M355_LIB=...] python tools/diag_intra.py [depths...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from libde265_amd import capi, synth, worklist
lib = capi.Library(); ctx = capi.Context(lib, 0)
depths = [int(a) for a in sys.argv[1:]] or [1]
for cu, cbf in ((0, 60), (3, 60), (3, 0), (4, 60), (5, 60), (6, 60)):
    cfg = dict(synth.CONFIGS["c2_1080p_intra"], fixed_cu_log2=cu, cbf_pct=cbf)
    pic = synth.picture(**cfg); pp = pic.pp[0]
    pic.ref_frames = [-1] * worklist.MAX_REF_FRAMES
    hs = []
    for _ in range(max(depths)):
        pic.dst_frame = ctx.frame_create_for(pp); hs.append((ctx.upload(pic), pic.dst_frame))
    ctx.wait()
    line = "cu_log2=%d cbf=%d%% blocks=%d:" % (cu, cbf, len(pic.ibs))
    for d in depths:
        ctx.set_pipeline_depth(d)
        for i in range(2 * d): ctx.decode_resident(hs[i % d][0])
        ctx.wait(); ctx.timing_reset()
        n = 10 * d
        t0 = time.perf_counter()
        for i in range(n): ctx.decode_resident(hs[i % d][0])
        ctx.wait(); dt = (time.perf_counter() - t0) / n
        _, tot, st = ctx.timing_collect()
        line += "  d%d: %.3f ms/pic (intra stage %.3f)" % (d, 1e3 * dt, st["intra"])
    ctx.set_pipeline_depth(1)
    print(line, flush=True)
    for h, f in hs:
        ctx.release(h); ctx.frame_destroy(f)
