"""The N>1 path on CPU: world_size-2 (and 3) torch.distributed jobs over gloo, one process per rank,
each running the product kernels under the SIMT interpreter on ITS tiles and exchanging tile-boundary
halos / finished tiles through the same ShardedDecoder the GPU job uses over RCCL."""
import json
import os
import socket
import subprocess
import sys

import pytest

from test_emu_picture import EMU_SO, emu_lib  # noqa: F401

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


@pytest.mark.parametrize("world,depth,halo", [(2, 1, "p2p"), (3, 1, "p2p"), (2, 3, "p2p"), (3, 1, "allreduce"), (4, 2, "p2p")])
def test_gloo_sharded_decode(emu_lib, oracle, world, depth, halo):  # noqa: F811
    cases = [dict(width=256, height=192, bit_depth=8, seed=51, tile_cols=2, tile_rows=2),
             dict(width=192, height=128, bit_depth=10, seed=52, tile_cols=3, tile_rows=1),
             dict(width=256, height=128, bit_depth=8, seed=53, tile_cols=4, tile_rows=2)]     # world 4: ranks that are not neighbours exchange nothing
    env = dict(os.environ, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.join(ROOT, "tests", "shard_worker.py"), "gloo", EMU_SO, json.dumps(cases), str(depth), halo]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "SHARD_WORKER_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
