"""One rank of the built-in RCCL transport test (tests/test_gpu_rccl.py): two processes share the ONE GPU of the box.
usage: rccl_worker.py <rank> <nranks> <id file> <result file>
Rank 0 writes the RCCL unique id; every rank creates a communicator on device 0 (m355_shard_rccl_init), runs the transport's
self-test (real bytes through ncclSend / ncclRecv / ncclAllGather + k_halo_add) and then decodes a 2x1-tile picture tile-sharded
(m355_decode_sharded: halo exchanges X0-X2 and the tile all-gather X3 over RCCL), comparing every frame with the oracle."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    rank, nranks, idfile, outfile = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    res = {"rank": rank, "stage": "start"}
    try:
        import ctypes
        from libde265_amd import capi, shard, worklist
        from oracle_py import Oracle
        from synth_util import make_case, oracle_decode, assert_planes_equal
        lib = capi.Library()
        ctx = capi.Context(lib, 0)
        if rank == 0:
            uid = lib.rccl_unique_id()
            with open(idfile + ".tmp", "wb") as f:
                f.write(bytes(uid))
            os.replace(idfile + ".tmp", idfile)
        else:
            t0 = time.time()
            while not os.path.exists(idfile):
                if time.time() - t0 > 45:
                    raise RuntimeError("no unique id from rank 0")
                time.sleep(0.05)
            uid = open(idfile, "rb").read()
        res["stage"] = "init"
        ctx.shard_rccl_init(uid, rank, nranks)
        res["stage"] = "selftest"
        ctx.shard_rccl_selftest(1 << 16)
        ctx.shard_rccl_selftest(77)
        res["selftest"] = "ok"
        res["stage"] = "decode"
        o = Oracle(ctypes.CDLL(os.path.join(ROOT, "oracle", "liboracle.so")))
        for case in (dict(width=416, height=240, bit_depth=8, seed=91, tile_cols=2, tile_rows=1),
                     dict(width=1280, height=720, bit_depth=10, seed=92, tile_cols=2, tile_rows=2)):
            pic, refs = make_case(**case)
            want = oracle_decode(o, pic, refs)
            pp = pic.pp[0]
            handles = []
            for planes in refs:
                f = ctx.frame_create_for(pp)
                ctx.frame_upload(f, planes)
                handles.append(f)
            sp = shard.shard_picture(pic, rank, nranks)           # this rank's tiles only (numpy, no torch in this process)
            sp.dst_frame = ctx.frame_create_for(pp)
            sp.ref_frames = [handles[i] if i < len(handles) else -1 for i in range(worklist.MAX_REF_FRAMES)]
            h = ctx.upload(sp)
            ctx.decode_sharded(h, True)                           # phases + X0..X3 issued by the library over RCCL
            ctx.decode_sharded(h, True)                           # (again into the same frame: the exchange buffers are reused)
            ctx.wait()
            assert_planes_equal(ctx.frame_download(sp.dst_frame), want, "rank %d of %d, %dx%d" % (rank, nranks, case["width"], case["height"]))
        res["decode"] = "ok"
        res["stage"] = "done"
        ctx.close()
    except Exception as e:  # noqa: BLE001
        res["error"] = repr(e)[:1500]
    with open(outfile, "w") as f:
        json.dump(res, f)


if __name__ == "__main__":
    main()
