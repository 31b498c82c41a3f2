#!/usr/bin/env python3
"""bench.py — decoded CTBs/s of the MI355X HEVC pixel-reconstruction path on synthetic work lists.

Contract (one JSON line on rank 0): a "step" = one pass of the hot path (inter prediction, residual,
intra wavefront, deblocking, SAO) over one synthetic picture of the named workload, with the work
lists and reference frames already resident in HBM when the timed region starts.  Default workload =
BASELINE.json configs[4] at one GPU: 8K (7680x4320) 10-bit, 4x2 tiles, full in-loop filter chain
("c5_8k10_8tiles").  value = CTB64/s over all ranks; for N>1 every rank decodes its
own picture stream (weak scaling, no data-path collective — see DESIGN.md §multi-GPU); the barrier and the
max-over-ranks clock of torch.distributed run over gloo (an idle RCCL communicator in the process costs the
kernels ~13 % here), RCCL is created for the additional tile-sharded measurement, which does exchange data.
The timed region runs with --pipeline-depth pictures in flight (default 3; the one-at-a-time time is reported
beside it).

Also reported:
  roofline      — the dominant kernel's algorithmic bytes per launch (SURVEY.md §8d accounting, computed
                  from the actual lists) / its average duration from HIP events on the library's stream,
                  against the 8 TB/s HBM3E peak;
  cpu_baseline  — the REAL reference functions (oracle/_ref, SSE/AVX tables where the reference has them) replaying a
                  bounded sample of the same recipe on this box's host cores, one picture stream per thread.
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
    ap.add_argument("--steps", type=int, default=1200, help="steps of THE timed region (default: ~0.5 s of device time at the default workload)")
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--repeats", type=int, default=0, help="the timed region of --steps steps is run this many times back to back, each bracketed by a "
                    "synchronisation; ms_per_step / value are the MEDIAN region, ms_per_step_spread the min / max (default: 25 for <= 200 steps, else 5)")
    ap.add_argument("--workload", default="c5_8k10_8tiles")
    ap.add_argument("--pipeline-depth", type=int, default=3, help="pictures in flight per GPU in the timed region (1..32)")
    ap.add_argument("--intra-batch", type=int, default=0, help="intra workloads: the timed region's steps go out in groups of this many pictures with ONE intra stage "
                    "(m355_decode_batch; needs --pipeline-depth >= the group); a step is still one whole picture")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--group-sync", type=int, default=0, help="DIAGNOSTIC ONLY: wait for the device after every N steps of the timed region (pictures start in phase-aligned groups)")
    ap.add_argument("--stages", type=int, default=31, help="DIAGNOSTIC ONLY: M355_STAGE_* mask (anything but 31 is not a valid benchmark)")
    ap.add_argument("--no-with-upload", action="store_true", help="skip the PCIe-inclusive legs (lists recorded into the pinned arena -> validation -> H2D -> decode, per step)")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the bitstream-level leg (synthetic 8K stream through the reference CLI on the reference library and on the glue library)")
    ap.add_argument("--no-dependent-chain", action="store_true", help="skip the leg in which every picture references the two decoded before it")
    ap.add_argument("--no-cold-refs", action="store_true", help="skip the leg whose reference frames rotate through four pairs (working set beyond the Infinity Cache)")
    ap.add_argument("--force-tile-shard", action="store_true", help="run the tile-sharded measurement even at world size 1 (plumbing check)")
    ap.add_argument("--group-ranks", type=int, default=0, help="N=1 only, diagnostic: additionally decode the picture tile-sharded over this many contexts of THIS process (m355_group_*), all on the one GPU")
    ap.add_argument("--no-verify", action="store_true", help="skip the frame checks of the legs against the CPU oracle (they run after the timed regions)")
    ap.add_argument("--no-tile-shard", action="store_true", help="N>1: skip the additional tile-sharded (one picture across all GPUs) measurement")
    args = ap.parse_args()
    # --gpus N without a launcher around us: become the launcher (one rank per GPU under torch.distributed.run, as the contract's
    # own command line does) instead of silently measuring one GPU and printing n_gpus 1
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import socket
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.stdout.flush(); sys.stderr.flush()
        os.execv(sys.executable, cmd)
    if args.gpus != int(os.environ.get("WORLD_SIZE", "1")):
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%s: launch one rank per GPU (python -m torch.distributed.run --nproc-per-node %d bench.py --gpus %d ...)"
                         % (args.gpus, os.environ.get("WORLD_SIZE", "1"), args.gpus, args.gpus))
    # stdout carries exactly ONE line (rank 0's JSON): the libraries underneath are chatty on fd 1 (RCCL's version banner at
    # NCCL_DEBUG=VERSION, gloo's "[Gloo] Rank ... is connected" line), so fd 1 is pointed at stderr for the run and the
    # result is written to the saved descriptor at the end.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1 or ("RANK" in os.environ and "MASTER_ADDR" in os.environ):   # launched by torch.distributed.run
        # RCCL prints a version banner on STDOUT at NCCL_DEBUG=VERSION and above; stdout carries exactly one JSON line here
        if not os.environ.get("M355_KEEP_NCCL_DEBUG"):
            os.environ.pop("NCCL_DEBUG", None)
        import torch
        import torch.distributed as dist
        if os.environ.get("M355_BENCH_SHARE_GPU"):   # plumbing check of the N>1 path on a box with fewer GPUs than ranks
            local_rank %= max(1, torch.cuda.device_count())
        # torch's own HIP runtime is brought up only where it carries data (an nccl bootstrap group, the torch / RCCL transports of the tile-sharded leg): a second
        # runtime's queues in the process cost the library's kernels 6 % by existing and 14-18 % with streams in use (profiles/r06_v5_rccl_idle_ab.txt: what
        # round 2 booked as "an idle RCCL communicator"); the default transport of the tile-sharded leg (ipc) needs neither torch's runtime nor RCCL
        torch_gpu = os.environ.get("M355_BENCH_BACKEND", "gloo") == "nccl" or os.environ.get("M355_SHARD_TRANSPORT", "ipc") != "ipc"
        if torch.cuda.is_available():
            if torch_gpu:
                torch.cuda.set_device(local_rank)
        else:
            # no GPU: the CPU tier's plumbing check of this very script (tests/test_bench_launch.py: M355_LIB = the SIMT-interpreter build,
            # every rank on "device" 0, exchanges over gloo) — never a measurement
            local_rank = 0
        # The replica data path has no collective; torch.distributed only provides the barrier and the max-over-ranks
        # clock.  Those run over gloo: merely INITIALISING an RCCL communicator in the process slows every kernel of the
        # library by ~13 % on this stack (measured: 0.535 vs 0.460 ms per picture with / without an idle RCCL group), which
        # would be charged to multi-GPU scaling although nothing is exchanged.  RCCL is created afterwards, as a second
        # group, for the part that does exchange data (the tile-sharded leg).
        dist.init_process_group(os.environ.get("M355_BENCH_BACKEND", "gloo"))

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
    pic.ref_frames = [refs[i] if i < len(refs) else -1 for i in range(worklist.MAX_REF_FRAMES)]
    # one resident copy of the lists per picture in flight, each with its own destination frame (a decoder never reconstructs
    # consecutive pictures into the same frame; with one shared frame a picture without SAO could not overlap its predecessor at all)
    handles, dsts = [], []
    for _ in range(max(1, args.pipeline_depth)):
        pic.dst_frame = ctx.frame_create_for(pp)
        dsts.append(pic.dst_frame)
        handles.append(ctx.upload(pic))
    handle = handles[0]
    # frame checks of the legs (outside every timed region): the MD5 of the leg's LAST destination frame, made on the device (m355_frame_hash) or —
    # with_transfers — over the planes that arrived on the host, is compared in emit() with the CPU oracle's decode of the same lists and references
    # (oracle/ = the checker, never the thing measured): {leg: (md5 per plane, what the oracle has to decode)}
    seen = {}
    std_refs = ("independent", tuple(cfg["seed"] + 17 * i for i in range(cfg["n_refs"])))
    ctx.wait()
    if args.stages != 31:
        ctx.set_stages(args.stages)

    # (1) one picture at a time (pipeline depth 1): clean per-stage device timings for the roofline figures
    for _ in range(args.warmup):
        ctx.decode_resident(handle)
    ctx.wait()
    side_steps = max(1, min(args.steps, 200))   # the legs beside the timed region run a bounded number of steps
    ctx.timing_reset()
    t0 = time.perf_counter()
    for _ in range(side_steps):
        ctx.decode_resident(handle)
    ctx.wait()
    dt_serial = (time.perf_counter() - t0) * args.steps / side_steps
    n_dec, avg_total_ms, stage_ms = ctx.timing_collect()
    # (2) THE timed region: the same K steps with --pipeline-depth pictures in flight (m355_set_pipeline_depth: consecutive decodes
    # go round the lanes; every step still runs the whole chain for one picture, into the next of `depth` destination frames —
    # reusing a frame waits for its previous decode through the frame events)
    ctx.set_pipeline_depth(args.pipeline_depth)
    batch = args.intra_batch if args.intra_batch > 1 else 0
    if batch and (batch > len(handles) or cfg.get("intra_pct", 0) != 100):
        raise SystemExit("--intra-batch needs an intra workload and --pipeline-depth >= the batch")

    def run_steps(n):
        if not batch:
            for i in range(n):
                ctx.decode_resident(handles[i % len(handles)])
                if args.group_sync and (i + 1) % args.group_sync == 0:
                    ctx.wait()
            return
        i = 0
        while i < n:                          # groups of `batch` pictures, one shared k_intra launch each
            b = min(batch, n - i)
            ctx.decode_batch([handles[(i + j) % len(handles)] for j in range(b)])
            i += b
    run_steps(args.warmup)
    ctx.wait()
    repeats = args.repeats if args.repeats > 0 else (25 if args.steps <= 200 else 5)
    regions, enq = [], []
    for _ in range(repeats):                  # R timed regions of exactly K steps each: the run carries its own noise bar
        if dist:
            import torch
            dist.barrier(); dev_sync(torch)
        t0 = time.perf_counter()
        run_steps(args.steps)
        enq.append(time.perf_counter() - t0)  # host time to enqueue the K steps (launches are asynchronous)
        ctx.wait()
        if dist:
            dev_sync(torch)
        dt_r = time.perf_counter() - t0
        if dist:
            t = torch.tensor([dt_r], device="cuda" if dist.get_backend() == "nccl" else "cpu", dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt_r = float(t.item())
        regions.append(dt_r)
    if dist:
        dist.barrier()
    order = sorted(range(repeats), key=lambda k: regions[k])
    dt = regions[order[repeats // 2]]                     # the median region is THE timed region
    t_enq = enq[order[repeats // 2]]
    spread = {"repeats": repeats, "min": 1e3 * regions[order[0]] / args.steps, "median": 1e3 * dt / args.steps, "max": 1e3 * regions[order[-1]] / args.steps,
              "p10": 1e3 * regions[order[repeats // 10]] / args.steps, "p90": 1e3 * regions[order[min(repeats - 1, (9 * repeats) // 10)]] / args.steps}
    if rank == 0 and args.stages == 31 and not args.no_verify:
        last = (args.warmup + repeats * args.steps - 1) % len(handles)
        seen["headline"] = (ctx.frame_hash(dsts[last], capi.HASH_MD5), std_refs)
    ctx.set_pipeline_depth(1)

    # (2b) the same timed region with the reference frames ROTATING through four distinct pairs: the headline's pictures all read the same
    # two frames (2 x 99.5 MB at C5: inside the 256 MiB Infinity Cache), a stream's pictures read different ones.  Four pairs = 796 MB at
    # C5: every picture's windows come from HBM, not from the last-level cache.
    cold = None
    if not args.no_cold_refs and rank == 0 and cfg["n_refs"] >= 1:
        nr, n_sets = cfg["n_refs"], 4
        sets = []
        for k in range(n_sets):
            fr = []
            for i in range(nr):
                f = ctx.frame_create_for(pp)
                ctx.frame_upload(f, synth.ref_planes(cfg["seed"] + 17 * i + 1009 * (k + 1), int(pp["width"]), int(pp["height"]),
                                                     int(pp["chroma_format_idc"]), int(pp["bit_depth_luma"])))
                fr.append(f)
            sets.append(fr)
        saved = (pic.dst_frame, pic.ref_frames)
        ch, cd = [], []
        for k in range(max(n_sets, args.pipeline_depth)):
            pic.dst_frame = ctx.frame_create_for(pp)
            cd.append(pic.dst_frame)
            pic.ref_frames = [sets[k % n_sets][i] if i < nr else -1 for i in range(worklist.MAX_REF_FRAMES)]
            ch.append(ctx.upload(pic))
        pic.dst_frame, pic.ref_frames = saved
        ctx.wait()
        ctx.set_pipeline_depth(args.pipeline_depth)
        for k in range(2 * len(ch)):
            ctx.decode_resident(ch[k % len(ch)])
        ctx.wait()
        t0 = time.perf_counter()
        for k in range(side_steps):
            ctx.decode_resident(ch[k % len(ch)])
        ctx.wait()
        dtk = time.perf_counter() - t0
        if args.stages == 31 and not args.no_verify:
            kl = (side_steps - 1) % len(ch)
            seen["rotating_references"] = (ctx.frame_hash(cd[kl], capi.HASH_MD5), ("independent", tuple(cfg["seed"] + 17 * i + 1009 * (kl % n_sets + 1) for i in range(nr))))
        ctx.set_pipeline_depth(1)
        for h2 in ch:
            ctx.release(h2)
        for f in cd + [f for fr in sets for f in fr]:
            ctx.frame_destroy(f)
        ref_bytes = n_sets * nr * sum(int(a) * int(b) for a, b in worklist.plane_dims(int(pp["width"]), int(pp["height"]), int(pp["chroma_format_idc"]))) * (1 if pp["bit_depth_luma"] <= 8 else 2)
        cold = {"value": side_steps * n_ctbs / dtk, "unit": "CTB64/s", "ms_per_step": 1e3 * dtk / side_steps, "steps": side_steps, "reference_sets": n_sets,
                "reference_bytes": int(ref_bytes),
                "note": "the timed region's loop with the reference frames rotating through %d distinct pairs (%.0f MB: beyond the 256 MiB Infinity Cache), %d pictures in flight" % (n_sets, ref_bytes / 1e6, args.pipeline_depth)}

    # (3) a dependent chain: picture k is predicted from the pictures k-1 and k-2 (the same lists, uploaded once per frame
    # assignment; frames rotate), decoded with the same number of pictures in flight — every decode waits for the SAO of its
    # references (per-frame events in the library).  What a real P/B chain sees, next to the independent-picture headline.
    chain = None
    if not args.no_dependent_chain and rank == 0 and cfg["n_refs"] >= 1:
        nr = cfg["n_refs"]
        frames = [ctx.frame_create_for(pp) for _ in range(nr + 1)]
        for i in range(nr):
            ctx.frame_upload(frames[i + 1], synth.ref_planes(cfg["seed"] + 17 * i, int(pp["width"]), int(pp["height"]),
                                                             int(pp["chroma_format_idc"]), int(pp["bit_depth_luma"])))
        hs = []
        saved = (pic.dst_frame, pic.ref_frames)
        for i in range(nr + 1):
            pic.dst_frame = frames[i]
            pic.ref_frames = [frames[(i + 1 + k) % (nr + 1)] if k < nr else -1 for k in range(worklist.MAX_REF_FRAMES)]
            hs.append(ctx.upload(pic))
        pic.dst_frame, pic.ref_frames = saved
        ctx.wait()
        ctx.set_pipeline_depth(args.pipeline_depth)
        for k in range(2 * (nr + 1)):
            ctx.decode_resident(hs[k % (nr + 1)])
        ctx.wait()
        t0 = time.perf_counter()
        for k in range(side_steps):
            ctx.decode_resident(hs[k % (nr + 1)])
        ctx.wait()
        dtc = time.perf_counter() - t0
        if args.stages == 31 and not args.no_verify:
            # the chain's frame check: the start frames again, then nr + 2 decodes of the chain with nothing waited for in between (every frame written
            # at least once, the last picture predicted from two decoded ones) — the schedule of the timed loop, on a history short enough for the oracle
            ctx.wait()
            for i in range(nr):
                ctx.frame_upload(frames[i + 1], synth.ref_planes(cfg["seed"] + 17 * i, int(pp["width"]), int(pp["height"]),
                                                                 int(pp["chroma_format_idc"]), int(pp["bit_depth_luma"])))
            n_chk = nr + 2
            for k in range(n_chk):
                ctx.decode_resident(hs[k % (nr + 1)])
            ctx.wait()
            seen["dependent_chain"] = (ctx.frame_hash(frames[(n_chk - 1) % (nr + 1)], capi.HASH_MD5), ("chain", nr, n_chk))
        ctx.set_pipeline_depth(1)
        for h2 in hs:
            ctx.release(h2)
        for f in frames:
            ctx.frame_destroy(f)
        chain = {"value": side_steps * n_ctbs / dtc, "unit": "CTB64/s", "ms_per_step": 1e3 * dtc / side_steps, "steps": side_steps,
                 "note": "each picture references the %d decoded before it (decode waits for their SAO), %d lanes" % (nr, args.pipeline_depth)}

    # (4) PCIe-inclusive: the lists are written into the library's pinned arena (m355_arena_begin; libm355synth copies them
    # there with 16 threads, standing in for the parser's recorder threads), then validated, scheduled, copied to the device
    # and decoded, every step.  "submit_only": the same with the lists already lying in the arenas (three rotate).
    with_upload = None
    if not args.no_with_upload and rank == 0:
        ctx.set_pipeline_depth(args.pipeline_depth)
        st = {}
        for _ in range(13):                  # the rotating arenas (three) are allocated on first use
            ctx.submit_in_place(pic, state=st)
        ctx.wait()
        up_steps = max(side_steps, 100)      # (a host-side rate: 20 steps are 20 ms of wall clock, too few to be stable)
        t0 = time.perf_counter()
        for _ in range(up_steps):
            ctx.submit_in_place(pic, state=st)
        ctx.wait()
        dtu = time.perf_counter() - t0
        t0 = time.perf_counter()
        for _ in range(up_steps):
            ctx.submit_in_place(pic, state=st, refill=False)
        ctx.wait()
        dts = time.perf_counter() - t0
        if args.stages == 31 and not args.no_verify:
            seen["with_upload"] = (ctx.frame_hash(pic.dst_frame, capi.HASH_MD5), std_refs)
        # (4b) "with transfers" (SURVEY.md 8d): the H2D of the lists as above AND the D2H of every output frame — each decode goes
        # into one of `depth` frames, its download into pinned planes starts behind it (m355_frame_download_async: a copy engine
        # beside the kernels), and a frame is decoded into again only after its previous download has landed
        depth_t = max(1, args.pipeline_depth)
        tframes = [ctx.frame_create_for(pp) for _ in range(depth_t)]
        tplanes = [ctx.pinned_planes(f) for f in tframes]
        saved_dst = pic.dst_frame
        frame_bytes = sum(a.nbytes for a in tplanes[0][2])

        def transfer_steps(n):
            for k in range(n):
                f = tframes[k % depth_t]
                if k >= depth_t:
                    ctx.frame_download_wait(f)
                pic.dst_frame = f
                ctx.submit_in_place(pic, state=st, refill=False)
                ctx.frame_download_start(f, tplanes[k % depth_t])
            for f in tframes[:min(n, depth_t)]:
                ctx.frame_download_wait(f)
        transfer_steps(2 * depth_t)
        ctx.wait()
        t0 = time.perf_counter()
        transfer_steps(up_steps)
        ctx.wait()
        dtt = time.perf_counter() - t0
        if args.stages == 31 and not args.no_verify:
            import hashlib
            seen["with_transfers"] = ([hashlib.md5(a.tobytes()).digest() for a in tplanes[(up_steps - 1) % depth_t][2]], std_refs)     # (what landed in the pinned planes)
        pic.dst_frame = saved_dst
        for tp in tplanes:
            ctx.pinned_free(tp)
        for f in tframes:
            ctx.frame_destroy(f)
        for _ in range(3):
            ctx.submit(pic)
        ctx.wait()
        n_copy = max(1, up_steps // 4)
        t0 = time.perf_counter()
        for _ in range(n_copy):
            ctx.submit(pic)                   # the copying entry (lists anywhere in host memory): marshals in Python every step
        ctx.wait()
        dtcopy = time.perf_counter() - t0
        ctx.set_pipeline_depth(1)
        c_pic, keep = pic.to_c()
        nbytes = sum(a.nbytes for a in keep)
        with_upload = {"value": up_steps * n_ctbs / dtu, "unit": "CTB64/s", "ms_per_step": 1e3 * dtu / up_steps, "steps": up_steps,
                       "list_bytes_per_picture": int(nbytes),
                       "submit_only": {"value": up_steps * n_ctbs / dts, "ms_per_step": 1e3 * dts / up_steps},
                       "copying_submit": {"value": n_copy * n_ctbs / dtcopy, "ms_per_step": 1e3 * dtcopy / n_copy},
                       "with_transfers": {"value": up_steps * n_ctbs / dtt, "ms_per_step": 1e3 * dtt / up_steps, "frame_bytes": int(frame_bytes),
                                          "d2h_GBps": frame_bytes * up_steps / dtt / 1e9,
                                          "note": "submit_only + the D2H of EVERY output frame into pinned planes (what a player that reads each picture gets)"},
                       "note": "per step: lists written into the pinned arena (16 host threads) + validation + schedules + H2D + decode, %d pictures in flight; submit_only = without the writing; copying_submit = m355_submit_picture on lists elsewhere in host memory (incl. the Python marshalling of this harness)" % args.pipeline_depth}
    emitted = []

    def emit(sharded):
        """rank 0: the ONE JSON line (once)"""
        if rank != 0 or emitted:
            return
        emitted.append(1)
        ab = synth.algorithmic_bytes(pic)
        launches = {"inter": 1, "residual": 1, "intra": 1, "deblock": 2, "sao": 1}   # kernel launches per stage and picture
        dom = max(("inter", "residual", "intra", "deblock", "sao"), key=lambda s: stage_ms[s])
        per_launch_ms = stage_ms[dom] / max(1, launches[dom])
        achieved = (ab[dom] / max(1, launches[dom])) / (per_launch_ms * 1e-3) / 1e9 if per_launch_ms > 0 else 0.0
        try:
            copy_gbs = ctx.measure_copy_rate(1 << 30, 9)          # this box's device-to-device copy rate (read + written bytes): the achievable ceiling
        except Exception:  # noqa: BLE001
            copy_gbs = None
        out = {
            "metric": "decoded CTBs/s", "value": world * args.steps * n_ctbs / dt, "unit": "CTB64/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
            "repeats": repeats, "ms_per_step_spread": spread,
            "ms_per_step_one_in_flight": 1e3 * dt_serial / args.steps, "pictures_in_flight": args.pipeline_depth, "intra_batch": batch,
            "host_enqueue_ms_per_step": 1e3 * t_enq / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8" if pp["bit_depth_luma"] <= 8 else "u16", "data": "synthetic",
            "fps": world * args.steps / dt,
            "config": {"workload": args.workload, "width": int(pp["width"]), "height": int(pp["height"]),
                       "bit_depth": int(pp["bit_depth_luma"]), "tiles": "%dx%d" % (cfg["tile_cols"], cfg["tile_rows"]),
                       "ctbs_per_picture": n_ctbs, "stages": "inter+residual+intra+deblock+sao" if args.stages == 31 else "DIAGNOSTIC mask %d" % args.stages, "parallelism": "pictures/%d" % world},
            "algorithmic_bytes_per_ctb": ab["total"] / n_ctbs,
            "pipeline_GBps": ab["total"] * args.steps / dt / 1e9,
            "stage_ms": {k: round(v, 4) for k, v in stage_ms.items()},
            "roofline": {"bound": "hbm", "kernel": "k_" + dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic(args.workload, "k_" + dom),
                         "launch_ms": per_launch_ms, "algorithmic_bytes_per_launch": ab[dom] / max(1, launches[dom]),
                         "pictures_in_flight": 1,     # launch_ms / stage_ms: one picture at a time; `value`: args.pipeline_depth in flight
                         "traffic_total": pmc_traffic_total(args.workload),
                         # the ceiling a copy kernel reaches on THIS box (m355_measure_copy_rate: 1 GiB read + 1 GiB written per launch) and the fraction of it
                         "copy_ceiling_GBps": copy_gbs, "frac_of_copy_ceiling": (achieved / copy_gbs) if copy_gbs else None},
        }
        verdicts = verify_legs(seen, cfg, pic, synth, worklist) if seen else {}
        if "headline" in verdicts:
            out["verified"] = verdicts["headline"]
        if with_upload is not None:
            if "with_upload" in verdicts:
                with_upload["verified"] = with_upload["submit_only"]["verified"] = verdicts["with_upload"]
            if "with_transfers" in verdicts:
                with_upload["with_transfers"]["verified"] = verdicts["with_transfers"]
        if chain is not None and "dependent_chain" in verdicts:
            chain["verified"] = verdicts["dependent_chain"]
        if cold is not None and "rotating_references" in verdicts:
            cold["verified"] = verdicts["rotating_references"]
        if verdicts:
            out["verified_against"] = "oracle/liboracle.so (CPU restatement; the checker, outside every timed region): MD5 per plane of each leg's last destination frame"
        if sharded is not None:
            out["tile_sharded"] = sharded
        if with_upload is not None:
            out["with_upload"] = with_upload
        if world == 1 and args.group_ranks:
            grp_leg = in_process_group_leg(args, lib, cfg, pic, synth, worklist, 1e3 * dt / args.steps)
            if grp_leg is not None:
                out["tile_sharded_in_process"] = grp_leg
        if chain is not None:
            out["dependent_chain"] = chain
        if cold is not None:
            out["rotating_references"] = cold
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(cfg, synth, worklist)
        if not args.no_end_to_end and world == 1:
            e2e = end_to_end(cfg)
            if e2e is not None:
                out["end_to_end"] = e2e
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())

    sharded = None
    if dist and not args.no_tile_shard and (world > 1 or args.force_tile_shard):
        # The tile-sharded leg is the only part that runs collectives on the GPUs (RCCL).  It must never cost the main line:
        # exceptions are reported inside the JSON, and if the leg does not come back within the limit (a wedged collective
        # cannot be interrupted from Python) every rank prints / exits on its own from a watchdog thread.
        import threading
        limit = float(os.environ.get("M355_BENCH_SHARD_TIMEOUT", "240"))

        def give_up():
            emit({"error": "tile-sharded leg did not finish within %.0f s; replica result above is unaffected" % limit})
            os._exit(0)
        wd = threading.Timer(limit, give_up)
        wd.daemon = True
        wd.start()
        sharded = tile_sharded_leg(args, dist, torch, lib, local_rank, synth, worklist)
        wd.cancel()
    emit(sharded)
    ctx.close()
    if dist:
        dist.destroy_process_group()


def verify_legs(seen, cfg, pic, synth, worklist):
    """{leg: True / False} — the CPU oracle decodes what each leg's last destination frame must hold (seen[leg] = (MD5 per plane, recipe)) and the
    digests are compared.  Test infrastructure used as the CHECKER, after every timed region is over; {} when oracle/liboracle.so is not in the tree."""
    import hashlib
    so = os.path.join(ROOT, "oracle", "liboracle.so")
    if not os.path.exists(so):
        return {}
    try:
        from oracle_py import Oracle
        o = Oracle(ctypes.CDLL(so))
        pp = pic.pp[0]
        geo = (int(pp["width"]), int(pp["height"]), int(pp["chroma_format_idc"]), int(pp["bit_depth_luma"]))
        saved = pic.ref_frames
        cache, out = {}, {}

        def digest(fr):
            return [hashlib.md5(a.tobytes()).digest() for a in o.frame_planes(fr)]
        for leg, (got, recipe) in seen.items():
            if recipe not in cache:
                if recipe[0] == "independent":          # one decode from reference planes of the given seeds
                    refs = {}
                    for i, sd in enumerate(recipe[1]):
                        refs[i] = o.frame_new(pp); o.frame_set_planes(refs[i], synth.ref_planes(sd, *geo))
                    dst = o.frame_new(pp)
                    pic.ref_frames = [i if i < len(refs) else -1 for i in range(worklist.MAX_REF_FRAMES)]
                    assert o.decode(pic, dst, refs) == 0
                    cache[recipe] = digest(dst)
                    for f in list(refs.values()) + [dst]:
                        o.frame_free(f)
                else:                                   # ("chain", nr, n): bench.py's dependent chain from its start frames, n decodes
                    nr, n = recipe[1], recipe[2]
                    fr = [o.frame_new(pp) for _ in range(nr + 1)]
                    for i in range(nr):
                        o.frame_set_planes(fr[i + 1], synth.ref_planes(cfg["seed"] + 17 * i, *geo))
                    pic.ref_frames = [k if k < nr else -1 for k in range(worklist.MAX_REF_FRAMES)]
                    for k in range(n):
                        i = k % (nr + 1)
                        assert o.decode(pic, fr[i], {j: fr[(i + 1 + j) % (nr + 1)] for j in range(nr)}) == 0
                    cache[recipe] = digest(fr[(n - 1) % (nr + 1)])
                    for f in fr:
                        o.frame_free(f)
            out[leg] = bool(list(got) == cache[recipe])
        pic.ref_frames = saved
        return out
    except Exception as e:  # noqa: BLE001
        return {"error": repr(e)[:200]}


def dev_sync(torch):
    if torch.cuda.is_available() and torch.cuda.is_initialized():     # (only where torch's runtime carries work: ctx.wait() has drained the library's)
        torch.cuda.synchronize()


def pmc_traffic_total(workload):
    """HBM bytes per PICTURE over all kernels of the pipeline: the sum of profiles/pmc_traffic.json (per-launch figures x launches
    per picture as recorded there); null when no PMC passes of this workload are committed."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))[workload]
    except Exception:  # noqa: BLE001
        return None
    tot_f = tot_w = 0
    n = max((v.get("launches_sampled", 0) for k, v in d.items() if isinstance(v, dict) and k == "k_inter"), default=0) or \
        max((v.get("launches_sampled", 0) for k, v in d.items() if isinstance(v, dict)), default=0)
    if not n:
        return None
    for k, v in d.items():
        if not isinstance(v, dict) or "fetch_bytes" not in v:
            continue
        per_pic = v.get("launches_sampled", n) / n        # launches of this kernel group per picture
        tot_f += v["fetch_bytes"] * per_pic
        tot_w += v["write_bytes"] * per_pic
    return {"bytes_per_picture": int(tot_f + tot_w), "fetch_bytes": int(tot_f), "write_bytes": int(tot_w), "source": d.get("_source")}


def pmc_traffic(workload, kernel):
    """HBM bytes per launch of `kernel` from the rocprofv3 PMC passes committed under profiles/ (bench.py cannot
    run the profiler around itself): FETCH_SIZE x2 (the gfx950 correction of MI355X_MICROARCH.md, HBM section)
    + WRITE_SIZE, both reported in KiB by rocprofv3; null when no profile of this workload is committed."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            t = json.load(f)
        e = t[workload][kernel]
        return {"bytes_per_launch": e["fetch_bytes"] + e["write_bytes"], "fetch_bytes": e["fetch_bytes"], "write_bytes": e["write_bytes"],
                "source": t[workload].get("_source")}
    except Exception:
        return None


def tile_sharded_leg(args, dist, torch, lib, local_rank, synth, worklist):
    """N>1 only: ONE picture of the workload sharded by tiles over all ranks — every rank
    reconstructs its tiles, three halo exchanges feed deblocking / SAO across the tile boundaries, an all-gather
    completes the picture on every GPU (it is the reference for later pictures).  Strong scaling of one picture,
    reported next to the replica throughput; a failure here never costs the main line."""
    try:
        from libde265_amd import capi, shard
        rank, world = dist.get_rank(), dist.get_world_size()
        grp = None
        on_gpu = torch.cuda.is_available()
        if not on_gpu:
            os.environ.setdefault("M355_SHARD_TRANSPORT", "torch")     # CPU tier: the exchanges over the job's gloo group
        elif os.environ.get("M355_SHARD_TRANSPORT", "ipc") not in ("rccl", "ipc"):
            grp = dist.new_group(backend="nccl") if dist.get_backend() != "nccl" else None     # torch's RCCL group for the exchanges
        cfg = dict(synth.CONFIGS[args.workload])
        pic = synth.picture(**cfg)                                 # the SAME picture on every rank
        pp = pic.pp[0]
        ctx = capi.Context(lib, local_rank)
        # transport of the exchanges: "rccl" (default) = issued by the library itself (m355_decode_sharded: phase loop in C++, neighbour
        # ncclSend / ncclRecv + ncclAllGather on the picture's stream; the unique id travels over the bootstrap group);
        # "torch" = the same loop calling back into torch.distributed; "python" = the phase loop in Python (round-2 path)
        # "ipc" (default): the library's interprocess transport — exported buffers, interprocess events, a shared-memory segment (csrc/runtime_ipc.hip): one node,
        # no collective library, no second HIP runtime in the process
        transport = os.environ.get("M355_SHARD_TRANSPORT", "ipc")
        rccl_id = None
        if transport == "rccl":
            idt = torch.zeros(128, dtype=torch.uint8)
            if rank == 0:
                idt = torch.frombuffer(bytearray(lib.rccl_unique_id()), dtype=torch.uint8).clone()
            dist.broadcast(idt, 0)
            rccl_id = bytes(idt.tolist())
        if transport == "ipc":
            dec = shard.ShardedDecoder(ctx, rank, world, ipc_name="bench_%s_%s" % (os.environ.get("MASTER_PORT", "0"), os.environ.get("TORCHELASTIC_RUN_ID", "0")))
        else:
            dec = shard.ShardedDecoder(ctx, rank, world, comm=shard.DistComm(grp) if transport != "rccl" else None, device="cuda:%d" % local_rank if on_gpu else "cpu",
                                       halo=os.environ.get("M355_SHARD_HALO", "p2p"), native=transport != "python", rccl_id=rccl_id)
        refs = []
        for i in range(cfg["n_refs"]):
            f = ctx.frame_create_for(pp)
            ctx.frame_upload(f, synth.ref_planes(cfg["seed"] + 17 * i, int(pp["width"]), int(pp["height"]),
                                                 int(pp["chroma_format_idc"]), int(pp["bit_depth_luma"])))
            refs.append(f)
        # pictures in flight, like the main line: `depth` copies of the picture's lists with their own destination frames go round the lanes
        # (as many as the unsharded line.  Round 2 gave the sharded picture — a longer chain: nine small pack / unpack kernels and the
        # exchanges between its phases — one lane more, 0.43 vs 0.47 ms at world 1 then; with today's kernels the fourth lane costs what it
        # costs the unsharded line: 0.489 ms with 4 in flight, 0.445 ms with 3, profiles/r04_af_variants_ring_sharded.txt)
        depth = max(1, args.pipeline_depth)
        if os.environ.get("M355_BENCH_SHARD_DEPTH"):      # (A/B of the sharded leg's pictures in flight)
            depth = max(1, int(os.environ["M355_BENCH_SHARD_DEPTH"]))
        ctx.set_pipeline_depth(depth)
        sp = shard.shard_picture(pic, rank, world)
        sp.ref_frames = [refs[i] if i < len(refs) else -1 for i in range(worklist.MAX_REF_FRAMES)]
        hs, dsts = [], []
        for _ in range(depth):
            sp.dst_frame = ctx.frame_create_for(pp)
            dsts.append(sp.dst_frame)
            hs.append(dec.upload(sp))
        h = hs[0]
        # one picture through the point-to-point halo path first: if the stack refuses it, every rank falls back to the all-reduce
        ok = 1
        try:
            dec.decode(h); ctx.wait(); dev_sync(torch)
        except Exception:  # noqa: BLE001
            ok = 0
        okt = torch.tensor([ok], dtype=torch.int32)
        dist.all_reduce(okt, op=dist.ReduceOp.MIN)
        if int(okt.item()) == 0:
            dec.halo = "allreduce"

        def timed(gather):
            for i in range(args.warmup):
                dec.decode(hs[i % depth], gather=gather)
            ctx.wait()
            dist.barrier(); dev_sync(torch)
            t0 = time.perf_counter()
            for i in range(args.steps):
                dec.decode(hs[i % depth], gather=gather)
            t_host = time.perf_counter() - t0
            ctx.wait()
            dev_sync(torch)
            t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item()), t_host
        dt, t_host = timed(True)
        dt_ng, _ = timed(False)             # the same without the finished-tile all-gather (a non-reference picture)
        # one picture at a time (latency of a single sharded picture)
        ctx.wait()
        dist.barrier(); dev_sync(torch)
        t0 = time.perf_counter()
        n1 = max(1, min(args.steps, 100))
        for _ in range(n1):
            dec.decode(h)
            ctx.wait()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt_one = float(t.item()) / n1
        # the exchanges on their own (device time per collective, the buffers of this picture)
        ex_ms = None
        if world > 1 and dec.native:
            ex_ms = [ctx.shard_time_exchange(h, k, 20) for k in range(4)]
        elif world > 1 and on_gpu:
            ex_ms = []
            for k in range(4):
                buf = dec.xbufs[h][k]
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                for _ in range(3):
                    dec._exchange(buf, k, h)
                torch.cuda.synchronize(); dist.barrier()
                e0.record()
                for _ in range(20):
                    dec._exchange(buf, k, h)
                e1.record(); torch.cuda.synchronize()
                ex_ms.append(e0.elapsed_time(e1) / 20)
        sp.dst_frame = dsts[0]
        dec.decode(h)                       # leave the complete picture behind for the check below
        ctx.wait()
        # every rank must now hold the identical, complete picture: compare a checksum of the frames
        planes = ctx.frame_download(sp.dst_frame)
        import zlib
        crc = zlib.crc32(b"".join(p.tobytes() for p in planes))
        c = torch.tensor([crc, -crc], dtype=torch.int64)
        dist.all_reduce(c, op=dist.ReduceOp.MAX)
        same = int(c[0].item()) == crc and int(c[1].item()) == -crc
        xb = [int(ctx.shard_xbuf_bytes(h, k)) for k in range(4)]
        dist.barrier()
        ctx.close()
        return {"value": args.steps * len(pic.ctbs) / dt, "unit": "CTB64/s", "ms_per_picture": 1e3 * dt / args.steps, "scaling": "strong",
                "pictures_in_flight": depth, "ms_per_picture_one_at_a_time": 1e3 * dt_one, "host_enqueue_ms_per_picture": 1e3 * t_host / args.steps,
                "non_reference_picture": {"value": args.steps * len(pic.ctbs) / dt_ng, "ms_per_picture": 1e3 * dt_ng / args.steps},
                "exchange_ms": ex_ms, "halo_exchange": dec.halo, "transport": transport,
                "tiles_per_rank": (cfg["tile_cols"] * cfg["tile_rows"]) / world, "frames_identical_on_all_ranks": bool(same),
                "exchange": {"halo_allreduce_bytes": xb[:3], "tile_allgather_bytes": xb[3]}}
    except Exception as e:  # noqa: BLE001
        import traceback
        traceback.print_exc(file=sys.stderr)
        return {"error": repr(e)[:300]}


def in_process_group_leg(args, lib, cfg, pic, synth, worklist, ms_unsharded):
    """One picture tile-sharded over the contexts of ONE process (m355_group_*: exchanges = event-ordered peer copies, no collective
    library).  On the one GPU of a bench box all ranks share the device, so this measures what sharding COSTS (cutting the picture,
    the pack / unpack kernels, the exchanges as device-to-device copies) — per-GPU work does not shrink; on a node, rank r = GPU r."""
    try:
        from libde265_amd import capi, shard
        n_tiles = cfg["tile_cols"] * cfg["tile_rows"]
        ranks = max(2, min(args.group_ranks, n_tiles))
        if n_tiles < 2:
            return None
        pp = pic.pp[0]
        depth = 3
        ctxs = [capi.Context(lib, 0) for _ in range(ranks)]
        grp = capi.Group(lib, ctxs)
        try:
            hs = []
            for r, ctx in enumerate(ctxs):
                ctx.set_pipeline_depth(depth)
                refs = []
                for i in range(cfg["n_refs"]):
                    f = ctx.frame_create_for(pp)
                    ctx.frame_upload(f, synth.ref_planes(cfg["seed"] + 17 * i, int(pp["width"]), int(pp["height"]), int(pp["chroma_format_idc"]), int(pp["bit_depth_luma"])))
                    refs.append(f)
                sp = shard.shard_picture(pic, r, ranks)
                sp.ref_frames = [refs[i] if i < len(refs) else -1 for i in range(worklist.MAX_REF_FRAMES)]
                per = []
                for _ in range(depth):
                    sp.dst_frame = ctx.frame_create_for(pp)
                    per.append(ctx.upload(sp))
                hs.append(per)
            steps = max(30, min(args.steps, 200))

            def timed(gather, in_flight):
                for k in range(6):
                    grp.decode([hs[r][k % depth] for r in range(ranks)], gather)
                grp.wait()
                t0 = time.perf_counter()
                for k in range(steps):
                    grp.decode([hs[r][k % depth] for r in range(ranks)], gather)
                    if not in_flight:
                        grp.wait()
                grp.wait()
                return 1e3 * (time.perf_counter() - t0) / steps
            ms_ref = timed(True, True)
            ms_nonref = timed(False, True)
            ms_one = timed(True, False)
        finally:
            grp.close()
            for ctx in ctxs:
                ctx.close()
        return {"ranks": ranks, "all_ranks_on_one_gpu": True, "pictures_in_flight": depth, "ms_per_picture": ms_ref,
                "ms_per_picture_non_reference": ms_nonref, "ms_per_picture_one_at_a_time": ms_one,
                "tile_gather_ms": max(0.0, ms_ref - ms_nonref), "overhead_vs_unsharded": (ms_ref / ms_unsharded) if ms_unsharded > 0 else None,
                "note": "one process, one context per rank, exchanges = hipMemcpyPeerAsync between the contexts ordered by events (m355_group_decode); "
                        "PLUMBING figure: the ranks share this box's one GPU and their 6 streams each the HIP runtime's 4 hardware queues, so the "
                        "kernels of different ranks mostly serialise (profiles/r04_m_group_one_gpu.txt) — on a node rank r has GPU r to itself"}
    except Exception as e:  # noqa: BLE001
        return {"error": repr(e)[:300]}


def end_to_end(cfg):
    """Bitstream level, through the reference's public API: a synthetic stream of the workload's geometry (oracle/_ref/streamgen:
    I + P/B pictures, the workload's tiles and bit depth) decoded by the reference CLI (dec265 -q -t N) linked with (a) the
    reference library — its SSE/AVX paths where it has them — and (b) glue/_build/libde265.so = the same parser with every pixel
    produced by the MI355X backend.  Both are bounded by the reference's CABAC parser (one thread per tile); the figure that
    isolates the backend is `value`.  None when the reference-side binaries are not in the tree."""
    import re
    import subprocess
    import tempfile
    gen = os.path.join(ROOT, "oracle", "_ref", "streamgen")
    ref = os.path.join(ROOT, "oracle", "_ref", "dec265")
    glue = os.path.join(ROOT, "glue", "_build", "dec265")
    if not all(os.path.exists(p) for p in (gen, ref, glue)):
        return None
    try:
        # a workload without tiles (C3, the stand-in for the 4K ra_main conformance stream) gets the random-access stream shape:
        # hierarchical-B groups of 8 decoded out of output order, slice-header reference picture sets, 4 active references per
        # list, a long-term picture, temporal MVP, sign data hiding, and WPP substreams parsed by 8 threads (oracle/ref_streamgen.cc)
        ra = cfg["tile_cols"] * cfg["tile_rows"] == 1
        frames = 17 if ra else 16
        features = (2048 | 4096 | 8192 | 16384 | 32768) if ra else 0
        with tempfile.TemporaryDirectory() as td:
            path = os.path.join(td, "s.h265")
            subprocess.run([gen, path, str(cfg["width"]), str(cfg["height"]), str(cfg["bit_depth"]), str(cfg["tile_cols"]), str(cfg["tile_rows"]),
                            str(frames), "77", "5", "1", "1", str(features)], check=True, timeout=300)
            size = os.path.getsize(path)
            threads = 8 if ra else max(1, min(32, cfg["tile_cols"] * cfg["tile_rows"]))   # the parser runs one thread per tile / per CTB row in flight

            def run(exe, output=False, nthreads=None, scalar=False):
                env = dict(os.environ, M355_PIPELINE_DEPTH="3")
                env.pop("M355_LIB", None)
                cmd = [exe, "-q", "-t", str(threads if nthreads is None else nthreads)] + (["-0"] if scalar else []) + (["-o", "/dev/null"] if output else []) + [path]
                r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
                m = re.search(r"nFrames decoded: (\d+) \(\d+x\d+ @\s*([0-9.]+) fps\)", r.stdout + r.stderr)
                return (int(m.group(1)), float(m.group(2))) if m else (0, 0.0)
            run(glue)                                                       # warm-up (device context, arenas)
            nr, fr = run(ref)
            ng, fg = run(glue)
            # the same with every output picture handed to the application (dec265 -o: de265_get_image_plane on each picture,
            # i.e. for the backend a device-to-host copy of every frame into the pinned planes)
            nro, fro = run(ref, True)
            ngo, fgo = run(glue, True)
            # the reference with its SIMD tables switched off (dec265 -0) and both decoders without worker threads (-t 0)
            _, frs = run(ref, scalar=True)
            _, fr0 = run(ref, nthreads=0)
            _, fr0s = run(ref, nthreads=0, scalar=True)
            _, fg0 = run(glue, nthreads=0)
        ctbs = ((cfg["width"] + 63) // 64) * ((cfg["height"] + 63) // 64)
        return {"stream": "%dx%d %d-bit, %dx%d tiles, %d pictures (%s), %d bytes; dec265 -q -t %d" %
                          (cfg["width"], cfg["height"], cfg["bit_depth"], cfg["tile_cols"], cfg["tile_rows"], frames,
                           "random access: hierarchical-B groups of 8, 4 references per list, long-term picture, TMVP, SDH, WPP" if ra else "I + P/B, 2 references", size, threads),
                "reference_fps": fr, "reference_ctb64_per_s": fr * ctbs, "mi355x_fps": fg, "mi355x_ctb64_per_s": fg * ctbs,
                "pictures": [nr, ng], "speedup": (fg / fr) if fr > 0 else None,
                "columns_fps": {"reference -t %d" % threads: fr, "reference -t %d -0 (scalar)" % threads: frs, "reference -t 0": fr0, "reference -t 0 -0 (scalar)": fr0s,
                                "mi355x -t %d" % threads: fg, "mi355x -t 0": fg0, "host_cpus": os.cpu_count()},
                "with_output": {"reference_fps": fro, "mi355x_fps": fgo, "pictures": [nro, ngo], "speedup": (fgo / fro) if fro > 0 else None,
                                "note": "dec265 -o /dev/null: every picture is taken by the application (the backend downloads each frame)"},
                "note": "both decoders spend most of each picture in the reference's CABAC / syntax parser (host, one thread per tile); the backend's own rate is `value` / `with_upload`"}
    except Exception as e:  # noqa: BLE001
        return {"error": repr(e)[:200]}


def cpu_baseline(cfg, synth, worklist):
    """CPU baseline on the host cores of this box, on a bounded sample (~20 s): pictures of the same recipe cropped to at most
    1920x1088 (510 CTB64), replayed by T NATIVE threads, one picture stream per thread (like frame-parallel decoding), each into its
    own planes — oracle/ref_replay.cc m355_ref_bench, no interpreter in the loop.
    kind "reference": the REAL reference functions with their SSE4.1/AVX2/AVX-512 tables driven over the same work lists
    (oracle/_ref/libde265_ref.so, built from /root/reference by build()); kind "port": the scalar C restatement oracle/hevc_oracle.c,
    when oracle/_ref is not available (timed from Python, one thread)."""
    import subprocess
    from synth_util import make_case
    small = dict(cfg, width=min(1920, cfg["width"]), height=min(1088, cfg["height"]), tile_cols=min(2, cfg["tile_cols"]), tile_rows=min(2, cfg["tile_rows"]))
    pic, refs = make_case(**small)
    ncpu = os.cpu_count() or 1
    ref_so = os.path.join(ROOT, "oracle", "_ref", "libde265_ref.so")
    if not os.path.exists(ref_so):
        from oracle_py import Oracle
        from synth_util import oracle_decode
        so = os.path.join(ROOT, "oracle", "liboracle.so")
        if not os.path.exists(so):
            subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "oracle"], check=True)
        o = Oracle(ctypes.CDLL(so))
        oracle_decode(o, pic, refs)
        t0 = time.perf_counter(); n1 = 0
        while time.perf_counter() - t0 < 10.0:
            oracle_decode(o, pic, refs); n1 += 1
        return {"value": n1 * len(pic.ctbs) / (time.perf_counter() - t0), "unit": "CTB64/s", "cores": 1, "kind": "port",
                "sample": "%d x (%dx%d %d-bit picture of the same recipe) through oracle/hevc_oracle.c (scalar), 1 thread" % (n1, small["width"], small["height"], small["bit_depth"]),
                "host_cores_available": ncpu}
    import numpy as np
    lib = ctypes.CDLL(ref_so)
    lib.m355_ref_bench.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.POINTER(ctypes.c_double)]
    lib.m355_ref_bench.restype = ctypes.c_long
    dt_ = np.uint8 if small["bit_depth"] <= 8 else np.uint16
    rp = (ctypes.c_void_p * (worklist.MAX_REF_FRAMES * 3))()
    keep = []
    for s_, planes in enumerate(refs):
        for c_, a in enumerate(planes):
            a = np.ascontiguousarray(a, dtype=dt_); keep.append(a); rp[s_ * 3 + c_] = a.ctypes.data
    pic.ref_frames = [i if i < len(refs) else -1 for i in range(worklist.MAX_REF_FRAMES)]
    cpic, k2 = pic.to_c()

    def run(threads, accel, seconds):
        el = ctypes.c_double(0)
        n = lib.m355_ref_bench(ctypes.byref(cpic), rp, worklist.STAGE_ALL, accel, threads, seconds, ctypes.byref(el))
        if n < 0:
            raise RuntimeError("m355_ref_bench failed: %d" % n)
        return n, n * len(pic.ctbs) / el.value
    run(1, 1, 0.2)                                    # warm-up (static tables, page faults)
    n1, r1 = run(1, 1, 2.5)
    _, r1s = run(1, 0, 2.5)
    T = max(1, min(32, ncpu))
    nT, rT = run(T, 1, 7.0)
    out = {"value": rT, "unit": "CTB64/s", "cores": T, "kind": "reference",
           "sample": "%d x (%dx%d %d-bit picture of the same recipe, %d CTB64) through the libde265 reference functions (SSE4.1/AVX2/AVX-512 tables where the reference has them; "
                     "8-bit only) via oracle/ref_replay.cc, %d native threads, one picture stream each" % (nT, small["width"], small["height"], small["bit_depth"], len(pic.ctbs), T),
           "host_cores_available": ncpu, "single_core_value": r1, "single_core_scalar_value": r1s}
    best = (rT, T)
    if ncpu > T:
        for Ta in sorted(set([min(ncpu, 64), min(ncpu, 128), ncpu])):
            if Ta <= T:
                continue
            nA, rA = run(Ta, 1, 4.0)
            out.setdefault("more_threads", []).append({"value": rA, "cores": Ta, "pictures": nA})
            if rA > best[0]:
                best = (rA, Ta)
    out["best"] = {"value": best[0], "cores": best[1]}    # the host's best figure over the thread counts tried
    if (small["width"], small["height"]) != (cfg["width"], cfg["height"]):
        # the workload's OWN picture (not the cropped sample) through the same reference functions: at least one whole replay per thread and count
        try:
            fpic, frefs = make_case(**cfg)
            frp = (ctypes.c_void_p * (worklist.MAX_REF_FRAMES * 3))()
            for s_, planes in enumerate(frefs):
                for c_, a in enumerate(planes):
                    a = np.ascontiguousarray(a, dtype=dt_); keep.append(a); frp[s_ * 3 + c_] = a.ctypes.data
            fpic.ref_frames = [i if i < len(frefs) else -1 for i in range(worklist.MAX_REF_FRAMES)]
            fcpic, fk2 = fpic.to_c()

            def frun(threads, seconds):
                el = ctypes.c_double(0)
                n = lib.m355_ref_bench(ctypes.byref(fcpic), frp, worklist.STAGE_ALL, 1, threads, seconds, ctypes.byref(el))
                if n < 0:
                    raise RuntimeError("m355_ref_bench failed: %d" % n)
                return {"value": n * len(fpic.ctbs) / el.value, "cores": threads, "pictures": int(n), "seconds": el.value}
            out["full_size"] = {"picture": "%dx%d %d-bit, %d CTB64 (the bench workload itself)" % (cfg["width"], cfg["height"], cfg["bit_depth"], len(fpic.ctbs)),
                                "unit": "CTB64/s", "by_threads": [frun(1, 0.5), frun(T, 3.0)] + ([frun(best[1], 3.0)] if best[1] != T else [])}
        except Exception as e:  # noqa: BLE001
            out["full_size"] = {"error": repr(e)[:200]}
    return out


if __name__ == "__main__":
    main()
