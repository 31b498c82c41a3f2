"""The interprocess tile-sharding transport (csrc/runtime_ipc.hip: exported exchange buffers, interprocess events, a shared-memory segment of sequence words) on
the CPU tier: the ranks are THREADS of this process, each with its own context on the SIMT interpreter (whose "interprocess" handles are plain pointers and whose
launches are synchronous) — what runs here for real is the transport's bookkeeping: the segment's creation / attachment, the export table and its generations, the
publish / await order of every exchange, the per-handle read-complete words of X3, the end-of-picture meeting, pictures in flight with handles going round, a
rank without tiles, a rank that fails.  The ordering on a device is the GPU tier's (tests/test_gpu_shard_ipc.py: rank PROCESSES sharing the box's GPU)."""
import os
import threading

import pytest

from shard_ipc_worker import run_rank
from test_emu_picture import emu_lib  # noqa: F401  (fixture)
from libde265_amd import capi

CASES = [
    ([dict(width=256, height=128, bit_depth=8, seed=41, tile_cols=2, tile_rows=1)], 2, 2),
    ([dict(width=256, height=192, bit_depth=8, seed=42, tile_cols=2, tile_rows=2), dict(width=256, height=192, bit_depth=10, seed=43, tile_cols=2, tile_rows=2)], 4, 3),
    ([dict(width=320, height=128, bit_depth=8, seed=44, tile_cols=3, tile_rows=1, lf_across_tiles=0)], 3, 1),
    ([dict(width=128, height=128, bit_depth=8, seed=47, tile_cols=2, tile_rows=1)], 4, 3),          # ranks 1 and 3 own nothing
    ([dict(width=384, height=192, bit_depth=8, seed=49, tile_cols=4, tile_rows=2)], 8, 2),
]


def run_threads(lib, oracle, cases, nranks, depth, env_host_sync=False):
    name = "emu%d_%d" % (os.getpid(), threading.get_ident() & 0xFFFF)
    out, errs = [None] * nranks, [None] * nranks

    def work(r):
        try:
            out[r] = run_rank(lib, oracle, r, nranks, name, cases, depth=depth)
        except Exception as e:  # noqa: BLE001
            errs[r] = e
    ts = [threading.Thread(target=work, args=(r,)) for r in range(nranks)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(600)
    assert not any(t.is_alive() for t in ts), "a rank hangs"
    return out, errs


@pytest.mark.parametrize("cases,nranks,depth", CASES, ids=lambda v: ("%dx%d_seed%d" % (v[0]["width"], v[0]["height"], v[0]["seed"])) if isinstance(v, list) else "n%d" % v)
def test_ipc_transport_ranks_as_threads(emu_lib, oracle, cases, nranks, depth):  # noqa: F811
    out, errs = run_threads(emu_lib, oracle, cases, nranks, depth)
    assert all(e is None for e in errs), errs
    assert all(n == depth * len(cases) for n in out), out


def test_ipc_transport_a_failing_rank_ends_the_others(emu_lib, oracle, monkeypatch):  # noqa: F811
    """rank 1 never arrives: rank 0's first wait for it ends with an error within the bound instead of spinning for ever"""
    monkeypatch.setenv("M355_IPC_TIMEOUT", "2")          # (read once per process: this test must be the first to reach a wait, or the default 30 s applies)
    from synth_util import make_case
    from libde265_amd import shard, worklist
    ctx = capi.Context(emu_lib, 0)
    try:
        dec = shard.ShardedDecoder(ctx, 0, 2, ipc_name="emu_fail_%d" % os.getpid())
        pic, refs = make_case(width=256, height=128, bit_depth=8, seed=41, tile_cols=2, tile_rows=1)
        pp = pic.pp[0]
        rf = []
        for planes in refs:
            f = ctx.frame_create_for(pp); ctx.frame_upload(f, planes); rf.append(f)
        sp = shard.shard_picture(pic, 0, 2)
        sp.ref_frames = [rf[i] if i < len(rf) else -1 for i in range(worklist.MAX_REF_FRAMES)]
        sp.dst_frame = ctx.frame_create_for(pp)
        h = dec.upload(sp)
        with pytest.raises(capi.M355Error):
            dec.decode(h)
    finally:
        ctx.close()
