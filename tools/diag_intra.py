#!/usr/bin/env python3
"""Intra stage time vs block structure (1080p all-intra, fixed CU size): separates the per-CTB fixed cost from the per-level cost."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from libde265_amd import capi, synth, worklist
lib = capi.Library(); ctx = capi.Context(lib, 0)
for cu in (6, 5, 4, 3, 0):
    for cbf in (0, 60):
        cfg = dict(synth.CONFIGS["c2_1080p_intra"], fixed_cu_log2=cu, cbf_pct=cbf)
        pic = synth.picture(**cfg); pp = pic.pp[0]
        pic.dst_frame = ctx.frame_create_for(pp); pic.ref_frames = [-1] * worklist.MAX_REF_FRAMES
        h = ctx.upload(pic); ctx.wait()
        for _ in range(2): ctx.decode_resident(h)
        ctx.wait(); ctx.timing_reset()
        for _ in range(5): ctx.decode_resident(h)
        n, tot, st = ctx.timing_collect()
        nib = len(pic.ibs)
        print("cu_log2=%d cbf=%d%%: intra %.3f ms, %d blocks (%.1f per CTB), %.1f us per CTB step (64 steps)" % (cu, cbf, st["intra"], nib, nib / 510.0, 1e3 * st["intra"] / 64))
        ctx.release(h); ctx.frame_destroy(pic.dst_frame)
