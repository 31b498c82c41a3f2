"""Drivers for tile-sharded decoding in the tests: `local_sharded_decode` runs N virtual ranks in ONE
process in lockstep (one context per rank on the same device / emulator) and performs the exchanges by
hand — the same phase calls, buffers and pack/unpack kernels as the multi-process path, without a
process group; `dist_sharded_decode` is the real torch.distributed path (gloo on CPU with the emulator
library, RCCL on GPUs)."""
import numpy as np

from libde265_amd import capi, shard, worklist


def _setup_rank(ctx, pic, refs, rank, nranks, device):
    pp = pic.pp[0]
    handles = []
    for planes in refs:
        f = ctx.frame_create_for(pp)
        ctx.frame_upload(f, planes)
        handles.append(f)
    dst = ctx.frame_create_for(pp)
    sp = shard.shard_picture(pic, rank, nranks)
    sp.dst_frame = dst
    sp.ref_frames = [handles[i] if i < len(handles) else -1 for i in range(worklist.MAX_REF_FRAMES)]
    return sp, dst


def local_sharded_decode(lib, pic, refs, nranks, device="cpu", stages=worklist.STAGE_ALL, repeat=1):
    """-> list (per rank) of the downloaded destination planes; every rank must hold the whole picture."""
    ctxs = [capi.Context(lib, 0) for _ in range(nranks)]
    try:
        decs, hs, dsts = [], [], []
        for r, ctx in enumerate(ctxs):
            ctx.set_stages(stages)
            d = shard.ShardedDecoder(ctx, r, nranks, comm=None, device=device, native=False)   # the phase loop by hand, in lockstep
            sp, dst = _setup_rank(ctx, pic, refs, r, nranks, device)
            decs.append(d); hs.append(d.upload(sp)); dsts.append(dst)
        for _ in range(repeat):
            for k in range(5):
                for d, h in zip(decs, hs):
                    d.run_phase(h, k)
                for ctx in ctxs:
                    ctx.wait()
                if k < 3:       # SUM all-reduce
                    total = sum(d.xbufs[h][k] for d, h in zip(decs, hs))
                    for d, h in zip(decs, hs):
                        d.xbufs[h][k].copy_(total)
                elif k == 3:    # all-gather of the rank slots
                    n = decs[0].xbufs[hs[0]][3].numel() // nranks
                    for r in range(nranks):
                        for d, h in zip(decs, hs):
                            d.xbufs[h][3][r * n:(r + 1) * n].copy_(decs[r].xbufs[hs[r]][3][r * n:(r + 1) * n])
                if device != "cpu":
                    import torch
                    torch.cuda.synchronize()
        return [ctx.frame_download(dst) for ctx, dst in zip(ctxs, dsts)]
    finally:
        for ctx in ctxs:
            ctx.close()


def dist_sharded_decode(lib, pic, refs, device="cpu", stages=worklist.STAGE_ALL, local_device=0, depth=1, halo="p2p"):
    """inside an initialised torch.distributed process group: this rank's destination planes.  depth > 1: pictures in
    flight — the picture is decoded into `depth` destination frames back to back, and once more into the first (a
    write-after-write on a frame whose previous decode may still be running); every frame must hold the same picture."""
    comm = shard.DistComm()
    ctx = capi.Context(lib, local_device)
    try:
        ctx.set_stages(stages)
        ctx.set_pipeline_depth(depth)
        d = shard.ShardedDecoder(ctx, comm.rank, comm.nranks, comm=comm, device=device, halo=halo)
        sp, dst = _setup_rank(ctx, pic, refs, comm.rank, comm.nranks, device)
        hs, dsts = [d.upload(sp)], [dst]
        for _ in range(depth - 1):
            sp.dst_frame = ctx.frame_create_for(pic.pp[0])
            dsts.append(sp.dst_frame)
            hs.append(d.upload(sp))
        for h in hs:
            d.decode(h)
        if depth > 1:
            d.decode(hs[0])
        ctx.wait()
        out = [ctx.frame_download(f) for f in dsts]
        for o in out[1:]:
            for a, b in zip(out[0], o):
                assert np.array_equal(a, b), "pictures in flight: destination frames differ"
        return out[0]
    finally:
        ctx.close()


def group_sharded_decode(lib, pic, refs, nranks, depth=1, gather=True, repeat=1, in_place=False):
    """The in-process group (m355_group_*): nranks contexts on device 0 decode one picture, the exchanges are copies between the
    contexts' buffers.  -> per rank the downloaded destination planes (gather: every rank holds the whole picture; else only its
    own tiles are meaningful).  depth > 1: `depth` copies of the lists / destination frames in flight.  in_place: every rank's lists
    are recorded into its handle's pinned arena (m355_picture_arena_begin -> m355_picture_replace: no host copy in the library), and
    recorded again into the same handle before every further round."""
    ctxs = [capi.Context(lib, 0) for _ in range(nranks)]
    grp = capi.Group(lib, ctxs)
    try:
        hs, dsts = [], []
        up = (lambda c, s_: c.upload_in_place(s_, slack=1.3)) if in_place else (lambda c, s_: c.upload(s_))
        shards = []
        for r, ctx in enumerate(ctxs):
            ctx.set_pipeline_depth(depth)
            per_h, per_d = [], []
            sp, dst = _setup_rank(ctx, pic, refs, r, nranks, "cpu")
            per_h.append(up(ctx, sp)); per_d.append(dst)
            for _ in range(depth - 1):
                sp.dst_frame = ctx.frame_create_for(pic.pp[0])
                per_d.append(sp.dst_frame)
                per_h.append(up(ctx, sp))
            hs.append(per_h); dsts.append(per_d); shards.append(sp)
        for it in range(repeat):
            for k in range(depth):
                if in_place and it:        # the handle's arena is recorded again (waits for its last decode only)
                    for r, ctx in enumerate(ctxs):
                        shards[r].dst_frame = dsts[r][k]
                        assert ctx.upload_in_place(shards[r], handle=hs[r][k], slack=1.0 + 0.2 * it) == hs[r][k]
                grp.decode([hs[r][k] for r in range(nranks)], gather)
        grp.wait()
        out = []
        for r, ctx in enumerate(ctxs):
            frames = [ctx.frame_download(f) for f in dsts[r]]
            for o in frames[1:]:
                for a, b in zip(frames[0], o):
                    assert np.array_equal(a, b), "pictures in flight: destination frames differ"
            out.append(frames[0])
        return out
    finally:
        grp.close()
        for ctx in ctxs:
            ctx.close()
