"""bench.py's own launch logic on the CPU tier: `python bench.py --gpus 2` with no launcher around it must start one rank per
"GPU" under torch.distributed.run itself (it used to measure one GPU and print n_gpus 1), run the replica line and the
tile-sharded leg, and print ONE JSON line with n_gpus 2.  The library here is the SIMT-interpreter build (M355_LIB), the picture a
256x192 one, the exchanges run over gloo: a plumbing check of the script, never a measurement."""
import json
import os
import subprocess
import sys

from test_emu_picture import EMU_SO, emu_lib  # noqa: F401

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(extra, env_extra=None):
    env = dict(os.environ, M355_LIB=EMU_SO, OMP_NUM_THREADS="1", M355_BENCH_SHARD_TIMEOUT="600")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "tiny_4tiles", "--steps", "2", "--warmup", "1", "--repeats", "1",
           "--no-cpu-baseline", "--no-end-to-end", "--no-with-upload", "--no-dependent-chain"] + extra
    return subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)


def test_bench_gpus_2_launches_itself(emu_lib):  # noqa: F811
    r = run_bench(["--gpus", "2"])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "pictures/2" and d["scaling"] == "weak"
    ts = d["tile_sharded"]
    assert "error" not in ts, ts
    assert ts["frames_identical_on_all_ranks"] is True and ts["tiles_per_rank"] == 2.0 and ts["scaling"] == "strong"


def test_bench_gpus_mismatch_fails_loudly(emu_lib):  # noqa: F811
    r = run_bench(["--gpus", "2"], env_extra={"WORLD_SIZE": "1", "RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stdout + r.stderr)
