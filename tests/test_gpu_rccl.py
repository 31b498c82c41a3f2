"""The built-in RCCL transport of the tile-sharded mode (csrc/runtime_shard.hip: m355_shard_rccl_init, rccl_halo_sum = grouped
ncclSend / ncclRecv per neighbour + k_halo_add, rccl_all_gather = ncclAllGather in place) on real hardware.  The pool's boxes
have ONE GPU, so (a) a communicator of one rank moves real bytes through every call (a grouped self-send), and (b) TWO rank
processes share the one GPU — if this RCCL build accepts two ranks on one device, the whole sharded decode (X0..X3 between two
processes) is compared with the oracle; if it refuses, the refusal text is what the test records and accepts (it is committed
under profiles/), never a silent skip of (a)."""
import json
import os
import subprocess
import sys

import pytest

from libde265_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_rccl_transport_moves_bytes_one_rank():
    lib = capi.Library()
    ctx = capi.Context(lib, 0)
    try:
        ctx.shard_rccl_init(lib.rccl_unique_id(), 0, 1)
        for words in (1, 63, 4096, 1 << 18):
            ctx.shard_rccl_selftest(words)
    finally:
        ctx.close()


@pytest.mark.skipif(os.environ.get("M355_TEST_RCCL_2RANKS") != "1",
                    reason="opt-in (M355_TEST_RCCL_2RANKS=1): the RCCL build of this image refuses a second rank on a device "
                           "(ncclCommInitRank fails, profiles/r04_e_rccl_two_ranks_one_gpu.json) and once hung in its bootstrap instead")
def test_rccl_two_ranks_share_the_gpu(tmp_path):
    idf = str(tmp_path / "rccl_id")
    outs = [str(tmp_path / ("rank%d.json" % r)) for r in range(2)]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", NCCL_DEBUG=os.environ.get("NCCL_DEBUG", "WARN"))
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "rccl_worker.py"), str(r), "2", idf, outs[r]],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env) for r in range(2)]
    logs = []
    for p in procs:
        try:
            logs.append(p.communicate(timeout=90)[0])
        except subprocess.TimeoutExpired:
            p.kill()
            logs.append("TIMEOUT\n" + p.communicate()[0])
    res = [json.load(open(o)) if os.path.exists(o) else {"error": "no result", "stage": "?"} for o in outs]
    report = {"results": res, "logs": [l[-3000:] for l in logs]}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(report, open(os.path.join(ROOT, "gpurun_out", "rccl_two_ranks_one_gpu.json"), "w"), indent=1)
    if all(r.get("stage") == "done" for r in res):
        return                                                   # two processes, one device: halo exchanges + all-gather, bit-exact
    # this RCCL build refuses (or cannot serve) two ranks on one device: that must happen at communicator creation / first use,
    # with an error — not as wrong data
    for r in res:
        assert r.get("decode") != "ok" or r.get("stage") == "done"
        assert "error" in r and r.get("stage") in ("init", "selftest"), report
    pytest.skip("RCCL does not serve two ranks on one device here: %s" % res[0].get("error", "")[:300])
