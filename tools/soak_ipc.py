#!/usr/bin/env python3
"""Soak of the INTERPROCESS tile-sharding transport (csrc/runtime_ipc.hip) on the hardware: groups of 2..4 rank PROCESSES sharing the GPU (tests/shard_ipc_worker.py: a
context per process, m355_shard_ipc_init, exported buffers mapped over HIP IPC, flag words in device memory), each group decoding a list of random tiled pictures
(tests/test_gpu_random.py random_case with a tile grid of at least as many tiles as ranks forced on it) with 1..3 handles in flight, every frame of every rank against
the oracle.  python tools/soak_ipc.py <first seed> <groups> <pictures per group> [groups at a time]  ->  a summary line; exit code 1 on any failure."""
import json
import multiprocessing as mp
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def group(args):
    first, g, per = args
    from synth_util import make_case
    from test_gpu_random import random_case
    rng = np.random.default_rng(77000 + first + g)
    nranks = int(rng.integers(2, 5))
    depth = int(rng.integers(1, 4))
    cases = []
    seed = first + g * per
    while len(cases) < per and seed < first + (g + 1) * per + 4 * per:
        case = random_case(seed)
        seed += 1
        ctbs_x, ctbs_y = -(-case["width"] >> case["log2_ctb"]), -(-case["height"] >> case["log2_ctb"])
        tc, tr = int(rng.integers(1, min(4, ctbs_x) + 1)), int(rng.integers(1, min(3, ctbs_y) + 1))
        if tc * tr < nranks:
            tc, tr = min(4, ctbs_x), min(3, ctbs_y)
        if tc * tr < nranks:
            continue
        case.update(tile_cols=tc, tile_rows=tr)
        try:
            make_case(**case)
        except RuntimeError:
            continue
        cases.append({k: int(v) for k, v in case.items()})
    name = "soak%d_%d" % (os.getpid(), g)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", M355_IPC_TIMEOUT=os.environ.get("M355_IPC_TIMEOUT", "120"))
    with tempfile.TemporaryDirectory() as td:
        procs = []
        for r in range(nranks):
            out = os.path.join(td, "r%d.json" % r)
            procs.append((out, subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "shard_ipc_worker.py"), str(r), str(nranks), name, out, json.dumps(cases), str(depth)],
                                                env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        res = []
        for out, p in procs:
            try:
                so, _ = p.communicate(timeout=int(os.environ.get("SOAK_IPC_RANK_TIMEOUT", "600")))
            except subprocess.TimeoutExpired:
                for _, q in procs:
                    q.kill()
                fin = [json.load(open(o2)) if os.path.exists(o2) else None for o2, _ in procs]     # (what the ranks that did return had to say)
                return nranks, depth, len(cases), ["a rank process hangs; the ranks' results: %r; first case %r" % (fin, cases[0])]
            res.append(json.load(open(out)) if os.path.exists(out) else {"ok": False, "error": "no result: " + (so or "")[-300:]})
    bad = [r.get("error", "frames %r" % r.get("frames")) for r in res if not r.get("ok") or r.get("frames") != depth * len(cases)]
    return nranks, depth, len(cases), bad


if __name__ == "__main__":
    first, groups, per = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    par = int(sys.argv[4]) if len(sys.argv) > 4 else 4
    t0 = time.time()
    with mp.get_context("spawn").Pool(par) as pool:
        only = os.environ.get("SOAK_IPC_ONLY")             # SOAK_IPC_ONLY=<g>: that group alone (its parameters follow from <first seed> and g)
        res = pool.map(group, [(first, g, per) for g in ([int(only)] if only else range(groups))])
    bad = [(r[0], r[1], b) for r in res for b in r[3]]
    print("soak_ipc: %d groups of 2..4 rank processes, %d pictures (%d frames of %d rank processes checked), %d FAIL, %.0f s, %d groups at a time"
          % (groups, sum(r[2] for r in res), sum(r[0] * r[1] * r[2] for r in res), sum(r[0] for r in res), len(bad), time.time() - t0, par))
    for b in bad[:10]:
        print("   ranks %d depth %d: %s" % b)
    sys.exit(1 if bad else 0)
