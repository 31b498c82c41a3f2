"""One rank of the interprocess transport (csrc/runtime_ipc.hip), as a function (the CPU tier runs the ranks as threads of one process on the SIMT interpreter)
and as a process (tests/test_gpu_shard_ipc.py: python shard_ipc_worker.py <rank> <nranks> <name> <result file> <case json> ... — the rank processes share the box's GPU).
Every rank: a context, m355_shard_ipc_init, its tiles' lists of every picture under `depth` handles going round, `rounds` sharded decodes of each with nothing
waited for in between (reference pictures: X0..X3; every third one a non-reference picture: no X3), the frames compared with the oracle's whole-picture decode."""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def run_rank(lib, oracle_cdll, rank, nranks, name, cases, depth=3, rounds=2, device=0):
    from oracle_py import Oracle
    from synth_util import assert_planes_equal, make_case, oracle_decode
    from libde265_amd import capi, shard, worklist
    o = Oracle(oracle_cdll)
    ctx = capi.Context(lib, device)
    try:
        dec = shard.ShardedDecoder(ctx, rank, nranks, ipc_name=name)
        ctx.set_pipeline_depth(depth)
        n_checked = 0
        diag = os.environ.get("M355_IPC_DIAG_DIR")           # (tools/soak_ipc.py: a line per picture — open file descriptors, what is being decoded — in <dir>/<name>_r<rank>.log)
        for ci, case in enumerate(cases):
            if diag:
                with open(os.path.join(diag, "%s_r%d.log" % (name, rank)), "a") as df:
                    df.write("case %d of %d: %d fds open, %r\n" % (ci, len(cases), len(os.listdir("/proc/self/fd")), case))
            pic, refs = make_case(**case)
            want = oracle_decode(o, pic, refs)
            pp = pic.pp[0]
            rf = []
            for planes in refs:
                f = ctx.frame_create_for(pp)
                ctx.frame_upload(f, planes)
                rf.append(f)
            sp = shard.shard_picture(pic, rank, nranks)
            sp.ref_frames = [rf[i] if i < len(rf) else -1 for i in range(worklist.MAX_REF_FRAMES)]
            hs, dsts = [], []
            for _ in range(depth):
                sp.dst_frame = ctx.frame_create_for(pp)
                dsts.append(sp.dst_frame)
                hs.append(dec.upload(sp))
            k = 0
            for _ in range(rounds):
                for i in range(depth):
                    dec.decode(hs[i], gather=(k % 3) != 2)
                    k += 1
            # the last decode of every handle as a reference picture, so that every frame is complete on every rank
            for i in range(depth):
                dec.decode(hs[i], gather=True)
            ctx.wait()
            for i in range(depth):
                assert_planes_equal(ctx.frame_download(dsts[i]), want, "rank %d of %d, frame %d, %r" % (rank, nranks, i, case))
                n_checked += 1
            for h in hs:
                dec.release(h)
            for f in rf + dsts:
                ctx.frame_destroy(f)
        return n_checked
    finally:
        ctx.close()


def main():
    rank, nranks, name, outfile = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    cases = json.loads(sys.argv[5])
    depth = int(sys.argv[6]) if len(sys.argv) > 6 else 3
    res = {"rank": rank, "ok": False}
    if os.environ.get("M355_IPC_DIAG_DIR"):
        import faulthandler
        faulthandler.dump_traceback_later(240, repeat=False, file=open(os.path.join(os.environ["M355_IPC_DIAG_DIR"], "%s_r%d.stack" % (name, rank)), "w"))
    try:
        from libde265_amd import capi
        lib = capi.Library()
        o = ctypes.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
        res["frames"] = run_rank(lib, o, rank, nranks, name, cases, depth=depth)
        res["ok"] = True
    except Exception as e:  # noqa: BLE001
        res["error"] = repr(e)[:600]
    with open(outfile, "w") as f:
        json.dump(res, f)


if __name__ == "__main__":
    main()
