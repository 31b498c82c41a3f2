"""Tile sharding on the GPU: the real HIP kernels, N virtual ranks on the one GPU of the
test box (one context per rank, lockstep phases, exchanges done with device-side tensor ops) vs the
oracle's whole-picture decode — small edge cases, then BASELINE's C4 (4K, 4 tiles, 4 ranks) and C5 (8K
10-bit, 8 tiles, 8 ranks) at full size; plus the torch.distributed/RCCL code path at world size 1."""
import json
import os
import subprocess
import sys

import pytest

from oracle_py import Oracle
from shard_util import group_sharded_decode, local_sharded_decode
from synth_util import assert_planes_equal, make_case, oracle_decode
from test_shard_emu import CASES
from test_shard_gloo import free_port
from libde265_amd import capi, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    L = capi.Library()
    assert L.device_count() >= 1
    return L


@pytest.mark.parametrize("case,nranks", CASES + [
    (dict(width=832, height=480, bit_depth=10, seed=61, tile_cols=3, tile_rows=2), 6),
    (dict(width=1280, height=720, bit_depth=8, seed=62, tile_cols=4, tile_rows=2, intra_pct=30), 3),
], ids=lambda v: ("%dx%d_seed%d" % (v["width"], v["height"], v["seed"])) if isinstance(v, dict) else "r%d" % v)
def test_sharded_small(lib, oracle, case, nranks):
    o = Oracle(oracle)
    pic, refs = make_case(**case)
    want = oracle_decode(o, pic, refs)
    for r, got in enumerate(local_sharded_decode(lib, pic, refs, nranks, device="cuda:0", repeat=2)):
        assert_planes_equal(got, want, "rank %d of %d" % (r, nranks))


@pytest.mark.parametrize("name,nranks", [("c4_4k_4tiles", 4), ("c5_8k10_8tiles", 8), ("c5_8k10_8tiles", 2)])
def test_sharded_baseline_configs(lib, oracle, name, nranks):
    o = Oracle(oracle)
    pic, refs = make_case(**synth.CONFIGS[name])
    want = oracle_decode(o, pic, refs)
    for r, got in enumerate(local_sharded_decode(lib, pic, refs, nranks, device="cuda:0")):
        assert_planes_equal(got, want, "%s rank %d of %d" % (name, r, nranks))


@pytest.mark.parametrize("depth", [1, 3])
def test_rccl_path_world_size_1(lib, oracle, depth):
    """the multi-process driver (ShardedDecoder + DistComm on the nccl backend, collectives ordered on the
    library's stream) with the one GPU this box has"""
    cases = [dict(width=416, height=240, bit_depth=8, seed=63, tile_cols=2, tile_rows=2),
             dict(width=1280, height=720, bit_depth=10, seed=64, tile_cols=3, tile_rows=2)]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.join(ROOT, "tests", "shard_worker.py"), "nccl", "default", json.dumps(cases), str(depth)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "SHARD_WORKER_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.parametrize("name,nranks,depth", [("c4_4k_4tiles", 4, 1), ("c5_8k10_8tiles", 8, 1), ("c5_8k10_8tiles", 4, 3), ("c3_4k_inter", 1, 2)])
def test_group_in_one_process_baseline_configs(lib, oracle, name, nranks, depth):
    """m355_group_*: one process, one context per rank (here all on the one GPU), exchanges = hipMemcpyPeerAsync between the
    contexts ordered by events; C4 / C5 at full size, also with three pictures in flight per rank"""
    o = Oracle(oracle)
    pic, refs = make_case(**synth.CONFIGS[name])
    want = oracle_decode(o, pic, refs)
    for r, got in enumerate(group_sharded_decode(lib, pic, refs, nranks, depth=depth, repeat=2)):
        assert_planes_equal(got, want, "%s group rank %d of %d" % (name, r, nranks))


@pytest.mark.parametrize("case,nranks", [(CASES[1][0], 4), (CASES[3][0], 3), (CASES[5][0], 3)],
                         ids=lambda v: ("%dx%d_seed%d" % (v["width"], v["height"], v["seed"])) if isinstance(v, dict) else "r%d" % v)
def test_group_in_one_process_small(lib, oracle, case, nranks):
    o = Oracle(oracle)
    pic, refs = make_case(**case)
    want = oracle_decode(o, pic, refs)
    for gather in (True, False):
        got_all = group_sharded_decode(lib, pic, refs, nranks, depth=2, gather=gather, repeat=2)
        if gather:
            for r, got in enumerate(got_all):
                assert_planes_equal(got, want, "group rank %d of %d" % (r, nranks))

