"""Worker of tests/test_shard_gloo.py: one rank of a torch.distributed job (gloo on CPU with the
SIMT-interpreter library, or nccl = RCCL with the product library on GPUs) decoding its tiles of a
synthetic picture and checking the COMPLETE picture it ends up with against the oracle."""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    import torch.distributed as dist
    from oracle_py import Oracle
    from shard_util import dist_sharded_decode
    from synth_util import assert_planes_equal, make_case, oracle_decode
    from libde265_amd import capi

    backend, libpath, cases = sys.argv[1], sys.argv[2], json.loads(sys.argv[3])
    depth = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    halo = sys.argv[5] if len(sys.argv) > 5 else "p2p"
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        dist.init_process_group("gloo")
    lib = capi.Library(libpath if libpath != "default" else None)
    o = Oracle(ctypes.CDLL(os.path.join(ROOT, "oracle", "liboracle.so")))
    for case in cases:
        pic, refs = make_case(**case)
        want = oracle_decode(o, pic, refs)
        got = dist_sharded_decode(lib, pic, refs, device="cuda:%d" % local_rank if backend == "nccl" else "cpu",
                                  local_device=local_rank if backend == "nccl" else 0, depth=depth, halo=halo)
        assert_planes_equal(got, want, "rank %d case %s" % (dist.get_rank(), case))
    dist.barrier()
    if dist.get_rank() == 0:
        print("SHARD_WORKER_OK %d ranks %d cases" % (dist.get_world_size(), len(cases)))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
