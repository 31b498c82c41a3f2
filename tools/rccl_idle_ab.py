#!/usr/bin/env python3
"""Why does an idle RCCL communicator in the process slow every kernel of the library by ~13 %?  World size 1, the bench workload, three pictures
in flight; every arm in a process of its own:
  base        no torch in the process
  torch       torch imported, its HIP runtime initialised, one tensor allocated (a SECOND HIP runtime in the process, no RCCL)
  streams     torch + four torch.cuda.Stream objects used once
  nccl        torch.distributed nccl group of one rank, one all_reduce (communicator created), then idle
  nccl_1ch    the same with NCCL_MIN_NCHANNELS = NCCL_MAX_NCHANNELS = 1
  nccl_gone   the group destroyed again before the measurement
  own_rccl    the library's own transport (librccl loaded with dlopen in the library's runtime, m355_shard_rccl_init of one rank), idle, no torch
usage: python tools/rccl_idle_ab.py [workload] [steps]   (prints one line per arm: ms per picture, p10 / p90 over 9 regions)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ARM = r'''
import os, sys, time
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
arm, workload, steps = sys.argv[1], sys.argv[2], int(sys.argv[3])
if arm in ("torch", "streams", "nccl", "nccl_1ch", "nccl_gone"):
    import torch
    torch.cuda.init(); t = torch.zeros(1 << 20, device="cuda"); torch.cuda.synchronize()
    if arm == "streams":
        ss = [torch.cuda.Stream() for _ in range(4)]
        for s in ss:
            with torch.cuda.stream(s):
                t.add_(1)
        torch.cuda.synchronize()
    if arm.startswith("nccl"):
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29611"); os.environ["RANK"] = "0"; os.environ["WORLD_SIZE"] = "1"
        dist.init_process_group("nccl", rank=0, world_size=1)
        dist.all_reduce(t); torch.cuda.synchronize()
        if arm == "nccl_gone":
            dist.destroy_process_group(); torch.cuda.synchronize()
from libde265_amd import capi, synth, worklist
lib = capi.Library(); ctx = capi.Context(lib, 0)
if arm == "own_rccl":
    ctx.shard_set(0, 1) if hasattr(ctx, "shard_set") else None
    ctx.shard_rccl_init(lib.rccl_unique_id(), 0, 1)
    ctx.shard_rccl_selftest(1 << 12)
cfg = dict(synth.CONFIGS[workload]); pic = synth.picture(**cfg); pp = pic.pp[0]
refs = []
for i in range(cfg["n_refs"]):
    f = ctx.frame_create_for(pp); ctx.frame_upload(f, synth.ref_planes(cfg["seed"] + 17 * i, int(pp["width"]), int(pp["height"]), int(pp["chroma_format_idc"]), int(pp["bit_depth_luma"]))); refs.append(f)
pic.ref_frames = [refs[i] if i < len(refs) else -1 for i in range(worklist.MAX_REF_FRAMES)]
hs = []
for _ in range(3):
    pic.dst_frame = ctx.frame_create_for(pp); hs.append(ctx.upload(pic))
ctx.wait(); ctx.set_pipeline_depth(3)
for i in range(10): ctx.decode_resident(hs[i %% 3])
ctx.wait()
r = []
for _ in range(9):
    t0 = time.perf_counter()
    for i in range(steps): ctx.decode_resident(hs[i %% 3])
    ctx.wait(); r.append(1e3 * (time.perf_counter() - t0) / steps)
r.sort()
print("RESULT %%s %%.4f %%.4f %%.4f" %% (arm, r[4], r[0], r[8]))
'''
workload = sys.argv[1] if len(sys.argv) > 1 else "c5_8k10_8tiles"
steps = sys.argv[2] if len(sys.argv) > 2 else "100"
for arm in (os.environ.get("M355_AB_ARMS", "base,torch,streams,nccl,nccl_1ch,nccl_gone,base").split(",")):
    env = dict(os.environ)
    if arm == "nccl_1ch":
        env.update(NCCL_MIN_NCHANNELS="1", NCCL_MAX_NCHANNELS="1")
    try:
        r = subprocess.run([sys.executable, "-c", ARM % dict(root=ROOT), arm, workload, steps], env=env, capture_output=True, text=True, timeout=300)
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")]
        print("%-10s %s" % (arm, line[0][7:] if line else "FAILED rc %d: %s" % (r.returncode, (r.stderr or r.stdout)[-300:].replace("\n", " | "))), flush=True)
    except subprocess.TimeoutExpired:
        print("%-10s TIMEOUT" % arm, flush=True)
