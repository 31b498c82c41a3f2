#!/usr/bin/env python3
"""Dependency-level counts of k_intra's chain for named workloads (runtime_upload.hip intra_schedule; host side only, interpreter library):
M355_INTRA_LEVEL_STATS=1 M355_INTRA_ONE_SIDED=0|1 python tools/intra_levels.py c2_1080p_intra c3_4k_inter c5_8k10_8tiles"""
import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from libde265_amd import capi, synth, worklist
lib = capi.Library('/root/repo/tests/simt_emu/_build/libde265_mi355x_emu.so')
ctx = capi.Context(lib, 0)
for name in sys.argv[1:]:
    cfg = dict(synth.CONFIGS[name]) if hasattr(synth, 'CONFIGS') else None
    pic = synth.picture(**cfg); pp = pic.pp[0]
    pic.dst_frame = ctx.frame_create_for(pp)
    refs = [ctx.frame_create_for(pp) for _ in range(pic.meta["cfg"]["n_refs"])]
    pic.ref_frames = [refs[i] if i < len(refs) else -1 for i in range(worklist.MAX_REF_FRAMES)]
    sys.stderr.write(name + ": "); sys.stderr.flush()
    h = ctx.upload(pic); ctx.release(h)
