#!/usr/bin/env python3
"""World-size-1 tile-sharded decode of the bench picture without a process group (a single rank exchanges nothing): what the
sharded machinery itself costs.  usage: [rocprofv3 --kernel-trace --stats --] python tools/prof_shard1.py [steps] [depth]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from libde265_amd import capi, shard, synth, worklist  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 3
torch.cuda.init()
lib = capi.Library()
cfg = dict(synth.CONFIGS["c5_8k10_8tiles"])
pic = synth.picture(**cfg)
pp = pic.pp[0]
ctx = capi.Context(lib, 0)
ctx.set_pipeline_depth(depth)
dec = shard.ShardedDecoder(ctx, 0, 1, comm=None, device="cuda:0")
refs = []
for i in range(cfg["n_refs"]):
    f = ctx.frame_create_for(pp)
    ctx.frame_upload(f, synth.ref_planes(cfg["seed"] + 17 * i, int(pp["width"]), int(pp["height"]), int(pp["chroma_format_idc"]), int(pp["bit_depth_luma"])))
    refs.append(f)
sp = shard.shard_picture(pic, 0, 1)
sp.ref_frames = [refs[i] if i < len(refs) else -1 for i in range(worklist.MAX_REF_FRAMES)]
hs = []
for _ in range(depth):
    sp.dst_frame = ctx.frame_create_for(pp)
    hs.append(dec.upload(sp))
for i in range(10):
    dec.decode(hs[i % depth], gather=False)
ctx.wait()
t0 = time.perf_counter()
for i in range(steps):
    dec.decode(hs[i % depth], gather=False)
t1 = time.perf_counter()
ctx.wait()
t2 = time.perf_counter()
print("sharded world 1, depth %d: %.4f ms/picture (host enqueue %.4f ms)" % (depth, 1e3 * (t2 - t0) / steps, 1e3 * (t1 - t0) / steps))
