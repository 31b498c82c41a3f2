#!/usr/bin/env python3
"""bench.py — decoded CTBs/s of the MI355X HEVC pixel-reconstruction path on synthetic work lists.

Contract (one JSON line on rank 0): a "step" = one pass of the hot path (inter prediction, residual,
intra wavefront, deblocking, SAO) over one synthetic picture of the named workload, with the work
lists and reference frames already resident in HBM when the timed region starts.  Default workload =
BASELINE.json configs[4] at one GPU: 8K (7680x4320) 10-bit, 4x2 tiles, full in-loop filter chain
("c5_8k10_8tiles").  value = CTB64/s over all ranks; for N>1 every rank decodes its
own picture stream (weak scaling, no data-path collective — see DESIGN.md §multi-GPU).

Also reported:
  roofline      — the dominant kernel's algorithmic bytes per launch (SURVEY.md §8d accounting, computed
                  from the actual lists) / its average duration from HIP events on the library's stream,
                  against the 8 TB/s HBM3E peak;
  cpu_baseline  — the CPU restatement (oracle, scalar port, 1 core) timed on a bounded sample of the
                  same recipe on this box's host cores.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec (6.3 TB/s measured copy ceiling)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c5_8k10_8tiles")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from libde265_amd import capi, synth, worklist
    lib = capi.Library()                      # raises if the HIP library is missing: no fallback
    ctx = capi.Context(lib, local_rank if world > 1 else 0)

    cfg = dict(synth.CONFIGS[args.workload])
    cfg["seed"] = (cfg["seed"] + 7919 * rank) & 0xFFFFFFFF       # every rank its own picture
    pic = synth.picture(**cfg)
    pp = pic.pp[0]
    n_ctbs = len(pic.ctbs)
    refs = []
    for i in range(cfg["n_refs"]):
        f = ctx.frame_create_for(pp)
        ctx.frame_upload(f, synth.ref_planes(cfg["seed"] + 17 * i, int(pp["width"]), int(pp["height"]),
                                             int(pp["chroma_format_idc"]), int(pp["bit_depth_luma"])))
        refs.append(f)
    pic.dst_frame = ctx.frame_create_for(pp)
    pic.ref_frames = [refs[i] if i < len(refs) else -1 for i in range(worklist.MAX_REF_FRAMES)]
    handle = ctx.upload(pic)
    ctx.wait()

    for _ in range(args.warmup):
        ctx.decode_resident(handle)
    ctx.wait()
    ctx.timing_reset()
    if dist:
        import torch
        dist.barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ctx.decode_resident(handle)
    ctx.wait()
    if dist:
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist:
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        dist.barrier()
    n_dec, avg_total_ms, stage_ms = ctx.timing_collect()

    if rank == 0:
        ab = synth.algorithmic_bytes(pic)
        launches = {"inter": 1, "residual": sum(1 for c in pic.rb_count if c), "intra": 1, "deblock": 2, "sao": 3}
        dom = max(("inter", "residual", "intra", "deblock", "sao"), key=lambda s: stage_ms[s])
        per_launch_ms = stage_ms[dom] / max(1, launches[dom])
        achieved = (ab[dom] / max(1, launches[dom])) / (per_launch_ms * 1e-3) / 1e9 if per_launch_ms > 0 else 0.0
        out = {
            "metric": "decoded CTBs/s", "value": world * args.steps * n_ctbs / dt, "unit": "CTB64/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8" if pp["bit_depth_luma"] <= 8 else "u16", "data": "synthetic",
            "fps": world * args.steps / dt,
            "config": {"workload": args.workload, "width": int(pp["width"]), "height": int(pp["height"]),
                       "bit_depth": int(pp["bit_depth_luma"]), "tiles": "%dx%d" % (cfg["tile_cols"], cfg["tile_rows"]),
                       "ctbs_per_picture": n_ctbs, "stages": "inter+residual+intra+deblock+sao", "parallelism": "pictures/%d" % world},
            "algorithmic_bytes_per_ctb": ab["total"] / n_ctbs,
            "pipeline_GBps": ab["total"] * args.steps / dt / 1e9,
            "stage_ms": {k: round(v, 4) for k, v in stage_ms.items()},
            "roofline": {"bound": "hbm", "kernel": "k_" + dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "launch_ms": per_launch_ms, "algorithmic_bytes_per_launch": ab[dom] / max(1, launches[dom])},
        }
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, synth, worklist)
        print(json.dumps(out))
    ctx.close()
    if dist:
        dist.destroy_process_group()


def cpu_baseline(cfg, synth, worklist):
    """The oracle (scalar C port of the reference's fallback path) on one core, on a bounded sample:
    pictures of the same recipe cropped to 1920x1088 (510 CTB64), repeated for >= ~10 s."""
    import subprocess
    from oracle_py import Oracle
    from synth_util import make_case, oracle_decode
    so = os.path.join(ROOT, "oracle", "liboracle.so")
    if not os.path.exists(so):
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "oracle"], check=True)
    o = Oracle(ctypes.CDLL(so))
    small = dict(cfg, width=1920, height=1088, tile_cols=min(2, cfg["tile_cols"]), tile_rows=min(2, cfg["tile_rows"]))
    pic, refs = make_case(**small)
    t0 = time.perf_counter(); n = 0
    while True:
        oracle_decode(o, pic, refs)
        n += 1
        if time.perf_counter() - t0 > 10.0 or n >= 40:
            break
    dt = time.perf_counter() - t0
    return {"value": n * len(pic.ctbs) / dt, "unit": "CTB64/s", "cores": 1, "kind": "port",
            "sample": "%d x (1920x1088 %d-bit picture of the same recipe, %d CTB64) through oracle/hevc_oracle.c" %
                      (n, small["bit_depth"], len(pic.ctbs)), "host_cores_available": os.cpu_count()}


if __name__ == "__main__":
    main()
