"""Tile-sharded decoding of one picture across GPUs — host side of SURVEY.md §8(e) over the C ABI
(`m355_shard_set`, `m355_decode_phase`, include/de265_mi355x.h "Tile sharding").

One process per GPU.  Rank r owns the tiles t (tile-scan order) with t*nranks//n_tiles == r, parses /
receives only those tiles' work lists (`shard_picture` cuts a whole-picture list down to that, for tests
and the benchmark; a real host parses just its tiles), and runs the five phases of a picture with an
exchange after each of the first four:

    X0  border-unit metadata + pre-deblock column strips   -> halo sum  (deblock.cc:191-209, :243-383)
    X1  post-vertical-pass row strips                       -> halo sum  (deblock.cc:919-939)
    X2  deblocked column + row strips (SAO ring)            -> halo sum  (sao.cc:83-88, :158-163)
    X3  finished tiles of the destination frame             -> all-gather      (reference for later pictures)

The buffers are torch tensors (device memory handed to RCCL through torch.distributed; CPU tensors when
the kernel-logic emulator of the test tier is the library).  Every element of X0..X2 is produced by
exactly one rank and zero elsewhere, so a sum completes them whatever the tile -> rank map is: by default a point-to-point
exchange with the ranks that own adjacent tiles (`neighbour_ranks`: one hop, the only producers of what a rank reads), or
an integer SUM all-reduce over all ranks (`halo="allreduce"`); traffic is a few hundred kB per picture, i.e. latency-bound on xGMI.  X3 is the real
volume (one picture per reference picture, 1/nranks of it sent by each rank).
"""
import numpy as np

from . import worklist


def owner_of_tile(t, n_tiles, nranks):
    """m355_shard_owner_of_tile: contiguous blocks of tiles in tile-scan order."""
    return 0 if nranks <= 1 else (t * nranks) // n_tiles


def ctb_owner_map(pp, nranks):
    """rank owning each CTB (raster order) for picture parameters `pp` (PIC_PARAMS record)."""
    cs = 1 << int(pp["log2_ctb_size"])
    w, h = (int(pp["width"]) + cs - 1) // cs, (int(pp["height"]) + cs - 1) // cs
    ntc, ntr = int(pp["num_tile_cols"]), int(pp["num_tile_rows"])
    out = np.zeros((h, w), np.int32)
    for ty in range(ntr):
        for tx in range(ntc):
            out[int(pp["row_bd"][ty]):int(pp["row_bd"][ty + 1]), int(pp["col_bd"][tx]):int(pp["col_bd"][tx + 1])] = \
                owner_of_tile(ty * ntc + tx, ntc * ntr, nranks)
    return out.reshape(-1), w


def shard_picture(pic, rank, nranks):
    """The part of a whole-picture work list that rank `rank` of `nranks` reconstructs: CUs, transform
    leaves, PBs, residual and intra blocks of its tiles (with their coefficients); slices[], ctbs[] (SAO
    parameters, slice indices), weights, PCM samples and the residual-buffer numbering stay picture-wide."""
    pp = pic.pp[0]
    owner, ctbw = ctb_owner_map(pp, nranks)
    l2 = int(pp["log2_ctb_size"])
    cf = int(pp["chroma_format_idc"])
    sw = 2 if cf in (1, 2) else 1
    sh = 2 if cf == 1 else 1

    def mine(x, y, cidx=None):
        x = x.astype(np.int64); y = y.astype(np.int64)
        if cidx is not None:
            x = np.where(cidx > 0, x * sw, x); y = np.where(cidx > 0, y * sh, y)
        return owner[(y >> l2) * ctbw + (x >> l2)] == rank

    out = worklist.Picture()
    out.pp = pic.pp.copy()
    out.dst_frame, out.ref_frames = pic.dst_frame, list(pic.ref_frames)
    out.slices, out.wts, out.pcm = pic.slices.copy(), pic.wts.copy(), pic.pcm.copy()
    out.scaling_factors, out.res_len, out.meta = pic.scaling_factors, pic.res_len, dict(pic.meta)
    out.cus = pic.cus[mine(pic.cus["x"], pic.cus["y"])] if len(pic.cus) else pic.cus.copy()
    out.tus = pic.tus[mine(pic.tus["x"], pic.tus["y"])] if len(pic.tus) else pic.tus.copy()
    out.pbs = pic.pbs[mine(pic.pbs["x"], pic.pbs["y"])] if len(pic.pbs) else pic.pbs.copy()
    # residual blocks: keep the size binning; compact the coefficient array
    if len(pic.rbs):
        keep = mine(pic.rbs["x"], pic.rbs["y"], pic.rbs["cidx"])
        bins = np.repeat(np.arange(4), pic.rb_count)
        out.rb_count = [int(np.count_nonzero(keep & (bins == s))) for s in range(4)]
        rbs = pic.rbs[keep].copy()
        n = rbs["ncoeff"].astype(np.int64)
        new_ofs = np.concatenate([[0], np.cumsum(n)[:-1]]) if len(rbs) else np.zeros(0, np.int64)
        idx = np.repeat(rbs["coeff_ofs"].astype(np.int64) - new_ofs, n) + np.arange(int(n.sum()))
        out.coeffs = pic.coeffs[idx] if len(idx) else np.zeros(0, np.dtype("<u4"))
        rbs["coeff_ofs"] = new_ofs.astype(np.uint32)
        out.rbs = rbs
    # intra blocks: whole CTBs are kept or dropped; renumber the CTBs' ranges
    ctbs = pic.ctbs.copy()
    keep_ctb = owner == rank
    if len(pic.ibs):
        order = np.argsort(ctbs["ib_start"], kind="stable")          # CTBs in decode (tile-scan) order own consecutive ranges
        ctb_of_ib = np.repeat(order, ctbs["ib_count"][order].astype(np.int64))
        keep_ib = keep_ctb[ctb_of_ib]
        out.ibs = pic.ibs[keep_ib]
        dropped_before = np.concatenate([[0], np.cumsum(~keep_ib)])[ctbs["ib_start"].astype(np.int64)]
        ctbs["ib_start"] = np.where(keep_ctb, ctbs["ib_start"].astype(np.int64) - dropped_before, 0).astype(np.uint32)
    ctbs["ib_count"] = np.where(keep_ctb, ctbs["ib_count"], 0)
    out.ctbs = ctbs
    return out


def neighbour_ranks(pp, rank, nranks):
    """Ranks that own a tile touching (edge or corner) a tile of `rank`: the only ranks whose halo elements `rank` consumes —
    deblocking reads across an edge, SAO the 1-sample ring incl. the diagonal corner.  Symmetric by construction."""
    ntc, ntr = int(pp["num_tile_cols"]), int(pp["num_tile_rows"])
    n = ntc * ntr
    out = set()
    for ty in range(ntr):
        for tx in range(ntc):
            if owner_of_tile(ty * ntc + tx, n, nranks) != rank:
                continue
            for dy in (-1, 0, 1):
                for dx in (-1, 0, 1):
                    x, y = tx + dx, ty + dy
                    if 0 <= x < ntc and 0 <= y < ntr:
                        q = owner_of_tile(y * ntc + x, n, nranks)
                        if q != rank:
                            out.add(q)
    return sorted(out)


class DistComm:
    """Exchanges over torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests)."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist, self.group = dist, group
        self.rank, self.nranks = dist.get_rank(group), dist.get_world_size(group)

    def all_reduce_sum(self, t):
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)

    def neighbour_sum(self, t, peers, scratch):
        """t += the same buffer of every rank in `peers` (point-to-point: one hop per neighbour instead of a ring / tree over all
        ranks; every element of a halo buffer has exactly one producer, and only adjacent ranks produce what this one reads).
        `scratch[i]` receives peer i's buffer.  Peers must list each other (neighbour_ranks is symmetric)."""
        if not peers:
            return
        d = self.dist
        g = self.group

        def gr(q):                                       # rank inside `group` -> global rank (P2POp takes global ranks)
            return d.get_global_rank(g, q) if g is not None else q
        ops = []
        for q, tmp in zip(peers, scratch):
            ops.append(d.P2POp(d.isend, t, gr(q), group=g))
            ops.append(d.P2POp(d.irecv, tmp, gr(q), group=g))
        for r in d.batch_isend_irecv(ops):
            r.wait()
        for tmp in scratch[:len(peers)]:
            t.add_(tmp)

    def all_gather_slots(self, t):
        n = t.numel() // self.nranks
        self.dist.all_gather_into_tensor(t, t[self.rank * n:(self.rank + 1) * n].clone(), group=self.group)


class _DevArray:
    """a raw device pointer as an object torch.as_tensor understands (__cuda_array_interface__)"""

    def __init__(self, ptr, n_words):
        self.__cuda_array_interface__ = {"shape": (n_words,), "typestr": "<i4", "data": (int(ptr), False), "version": 2}


class CallbackComm:
    """m355_comm callbacks over a DistComm (torch.distributed): what m355_decode_sharded calls between the phases when the
    transport is not the built-in RCCL one — the gloo CPU tests, or torch's own process group on the GPU."""

    def __init__(self, comm, device):
        import ctypes
        import torch
        from . import capi
        self.comm, self.torch, self.device = comm, torch, torch.device(device)
        self._streams = {}
        self.errors = []

        def tensor(ptr, nbytes):
            n = (nbytes + 3) // 4
            if self.device.type == "cuda":
                return torch.as_tensor(_DevArray(ptr, n), device=self.device)
            return torch.from_numpy(np.ctypeslib.as_array((ctypes.c_int32 * n).from_address(ptr)))

        def on_stream(stream_ptr, fn):
            if self.device.type != "cuda":
                return fn()
            st = self._streams.get(stream_ptr)
            if st is None:
                st = self._streams[stream_ptr] = torch.cuda.ExternalStream(stream_ptr, device=self.device)
            with torch.cuda.stream(st):
                return fn()

        def halo_sum(user, buf, nbytes, peers, n_peers, scratch, stream):
            try:
                pitch = (nbytes + 255) & ~255
                t = tensor(buf, nbytes)
                tmp = [tensor(scratch + pitch * i, nbytes) for i in range(n_peers)]
                on_stream(stream, lambda: self.comm.neighbour_sum(t, [peers[i] for i in range(n_peers)], tmp))
                return 0
            except Exception as e:  # noqa: BLE001
                self.errors.append(repr(e))
                return 1

        def all_gather(user, buf, slot_bytes, rank, nranks, stream):
            try:
                t = tensor(buf, slot_bytes * nranks)
                on_stream(stream, lambda: self.comm.all_gather_slots(t))
                return 0
            except Exception as e:  # noqa: BLE001
                self.errors.append(repr(e))
                return 1
        self._keep = (capi.HALO_SUM_FN(halo_sum), capi.ALL_GATHER_FN(all_gather))
        self.struct = capi.Comm(None, self._keep[0], self._keep[1])


class ShardedDecoder:
    """One rank's executor.  native=True (default): the phase loop and the exchange buffers live in the library
    (m355_decode_sharded) and the exchanges go through the m355_comm callbacks — the built-in RCCL transport
    (rccl_id = the unique id rank 0 made with rccl_unique_id and handed round), or `comm` (a DistComm) through CallbackComm.
    native=False: the phase loop in Python over m355_decode_phase (exchange buffers = torch tensors), kept as the cross-check."""

    def __init__(self, ctx, rank, nranks, comm=None, device="cuda", halo="p2p", native=True, rccl_id=None, ipc_name=None):
        self.ctx, self.rank, self.nranks, self.comm = ctx, rank, nranks, comm
        self.native = native and halo == "p2p"
        self.cb = None
        if ipc_name is not None:
            # the interprocess transport (csrc/runtime_ipc.hip: exported buffers + interprocess events + a shared-memory segment): no torch, no RCCL in
            # the data path — a second HIP runtime's streams in the process cost the library's kernels 6-18 % (profiles/r06_v5_rccl_idle_ab.txt)
            self.torch, self.device, self.native = None, None, True
            self.ctx.shard_ipc_init(ipc_name, rank, nranks)
            self.xbufs, self.halo, self._ptrs, self.peers, self.scratch, self._streams = {}, "p2p", {}, {}, {}, {}
            return
        import torch
        self.torch = torch
        self.device = torch.device(device)
        self.ctx.shard_set(rank, nranks)
        if self.native:
            if rccl_id is not None:
                self.ctx.shard_rccl_init(rccl_id, rank, nranks)       # ncclSend / ncclRecv / ncclAllGather issued by the library
            elif comm is not None and nranks > 1:
                self.cb = CallbackComm(comm, device)
                self.ctx.shard_set_comm(self.cb.struct)
        self.xbufs = {}
        self.halo = halo                     # "p2p": halos from the neighbour ranks only; "allreduce": SUM all-reduce over all ranks
        self._ptrs = {}
        self.peers = {}                      # picture handle -> neighbour ranks
        self.scratch = {}
        self._streams = {}
        # (torch's HIP runtime must have been initialised before the library's first HIP call in this
        # process — torch.cuda.init() / set_device() first — or torch finds no device.)

    def _lane_stream(self):
        """The library stream the picture's last phase was enqueued on (pictures in flight: one per lane) as a torch
        stream, so the collective that follows is ordered after that phase and before the picture's next one."""
        if self.device.type != "cuda":
            return None
        ptr = int(self.ctx.stream())
        st = self._streams.get(ptr)
        if st is None:
            st = self._streams[ptr] = self.torch.cuda.ExternalStream(ptr, device=self.device)
        return st

    def upload(self, pic_shard):
        h = self.ctx.upload(pic_shard)
        if self.native:
            return h
        self.xbufs[h] = [self.torch.zeros(max(1, self.ctx.shard_xbuf_bytes(h, k) // 4), dtype=self.torch.int32, device=self.device)
                         for k in range(4)]
        self._ptrs[h] = [t.data_ptr() for t in self.xbufs[h]]
        self.peers[h] = neighbour_ranks(pic_shard.pp[0], self.rank, self.nranks) if self.nranks > 1 else []
        if self.halo == "p2p" and self.peers[h]:
            big = max(self.xbufs[h][k].numel() for k in range(3))
            self.scratch[h] = [self.torch.empty(big, dtype=self.torch.int32, device=self.device) for _ in self.peers[h]]
        return h

    def release(self, h):
        self.ctx.release(h)
        self.xbufs.pop(h, None); self._ptrs.pop(h, None); self.peers.pop(h, None); self.scratch.pop(h, None)

    def run_phase(self, h, k):
        self.ctx.decode_phase(h, k, self._ptrs[h][k] if k < 4 else None)

    def exchange(self, h, k):
        if self.nranks == 1:
            return                           # a single rank owns every tile: nothing to exchange
        buf = self.xbufs[h][k]
        st = self._lane_stream()
        if st is not None:
            with self.torch.cuda.stream(st):
                self._exchange(buf, k, h)
        else:
            self.ctx.wait()                  # emulator: the "device" work is synchronous anyway
            self._exchange(buf, k, h)

    def _exchange(self, buf, k, h=None):
        if k < 3:
            if self.halo == "p2p" and h is not None:
                self.comm.neighbour_sum(buf, self.peers[h], [t[:buf.numel()] for t in self.scratch.get(h, [])])
            else:
                self.comm.all_reduce_sum(buf)
        else:
            self.comm.all_gather_slots(buf)

    def decode(self, h, gather=True):
        """All phases of one picture; asynchronous (GPU).  With ctx.set_pipeline_depth(n) consecutive pictures run on n lanes:
        the exchanges and filter phases of one picture overlap the prediction phase of the next wherever the frames they touch
        allow (every rank must then issue the same pictures in the same order — the collectives pair up by issue order).
        gather=False: a NON-reference picture — the finished tiles stay where they were decoded (each rank outputs its own
        tiles), no X3."""
        if self.native:
            self.ctx.decode_sharded(h, gather)
            if self.cb is not None and self.cb.errors:
                raise RuntimeError("exchange callback failed: %s" % self.cb.errors[0])
            return
        for k in range(5 if gather else 4):
            self.run_phase(h, k)
            if k < (4 if gather else 3):
                self.exchange(h, k)
