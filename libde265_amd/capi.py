"""ctypes binding of the C ABI declared in include/de265_mi355x.h (the product library
libde265_mi355x.so, built by csrc/Makefile for gfx950).  There is NO fallback: if the library is
missing or no HIP device is visible, loading / context creation raises."""
import ctypes
import os

import numpy as np

from . import worklist

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(_HERE, "libde265_mi355x.so")

ERRORS = {1: "M355_ERR_NO_DEVICE", 2: "M355_ERR_HIP", 3: "M355_ERR_INVALID", 4: "M355_ERR_NOMEM", 5: "M355_ERR_TIMEOUT", 6: "M355_ERR_BUSY", 7: "M355_ERR_STALE"}


HALO_SUM_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_int), ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p)
ALL_GATHER_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_void_p)


class Comm(ctypes.Structure):
    """m355_comm (include/de265_mi355x.h): the exchange callbacks of m355_decode_sharded"""
    _fields_ = [("user", ctypes.c_void_p), ("halo_sum", HALO_SUM_FN), ("all_gather", ALL_GATHER_FN)]


class M355Error(RuntimeError):
    def __init__(self, code, text):
        super().__init__("%s: %s" % (ERRORS.get(code, code), text))
        self.code = code


HASH_MD5, HASH_CRC, HASH_CHECKSUM = 0, 1, 2


class PictureHash(ctypes.Structure):
    """m355_picture_hash (include/de265_mi355x.h), the fields of sei_decoded_picture_hash (sei.h:64-69)"""
    _fields_ = [("md5", (ctypes.c_uint8 * 16) * 3), ("crc", ctypes.c_uint16 * 3), ("checksum", ctypes.c_uint32 * 3)]


class ArenaCaps(ctypes.Structure):
    """m355_arena_caps (include/de265_mi355x.h)"""
    _fields_ = [(n, ctypes.c_int32) for n in ("n_slices", "n_ctbs", "n_cus", "n_tus", "n_pbs", "n_wts", "n_ibs")] + \
               [("n_rbs", ctypes.c_int32 * 4), ("n_coeffs", ctypes.c_uint32), ("n_pcm", ctypes.c_uint32), ("scaling", ctypes.c_int32),
                ("rb_bin", ctypes.c_void_p * 4)]


class Library:
    """A loaded libde265_mi355x.so with typed entry points."""

    def __init__(self, path=None):
        # M355_LIB: an alternative BUILD of the same HIP library (kernel-variant experiments, tools/variants.sh)
        path = path or os.environ.get("M355_LIB") or DEFAULT_LIB
        if not os.path.exists(path):
            raise RuntimeError("MI355X backend library not found: %s (build it: make -C libde265_amd/csrc; "
                               "there is no CPU fallback)" % path)
        self.path = path
        L = self.lib = ctypes.CDLL(path)
        vp, i, cp = ctypes.c_void_p, ctypes.c_int, ctypes.c_char_p
        L.m355_last_error.restype = cp
        L.m355_version.restype = cp
        L.m355_device_count.restype = i
        L.m355_create.argtypes = [i, ctypes.POINTER(vp)]
        L.m355_destroy.argtypes = [vp]
        L.m355_destroy.restype = None
        L.m355_frame_create.argtypes = [vp, i, i, i, i, i]
        L.m355_frame_destroy.argtypes = [vp, i]
        L.m355_frame_upload.argtypes = [vp, i, i, vp, ctypes.c_ssize_t]
        L.m355_frame_download.argtypes = [vp, i, i, vp, ctypes.c_ssize_t]
        L.m355_frame_fill.argtypes = [vp, i, i, i]
        L.m355_measure_copy_rate.argtypes = [vp, ctypes.c_size_t, i, ctypes.POINTER(ctypes.c_double)]
        L.m355_arena_begin.argtypes = [vp, vp, vp]
        if hasattr(L, "m355_picture_arena_begin"):      # (absent from older builds loaded through M355_LIB for an A/B)
            L.m355_picture_arena_begin.argtypes = [vp, i, vp, vp, vp]
        L.m355_frame_download_async.argtypes = [vp, i, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_ssize_t)]
        L.m355_frame_download_wait.argtypes = [vp, i]
        L.m355_host_alloc.argtypes = [ctypes.c_size_t]
        L.m355_host_alloc.restype = vp
        L.m355_host_free.argtypes = [vp]
        L.m355_frame_hash.argtypes = [vp, i, i, vp]
        L.m355_submit_picture.argtypes = [vp, vp]
        L.m355_wait.argtypes = [vp]
        L.m355_last_serial.argtypes = [vp]
        L.m355_last_serial.restype = ctypes.c_ulonglong
        L.m355_decode_status.argtypes = [vp, ctypes.c_ulonglong]
        L.m355_picture_upload.argtypes = [vp, vp]
        L.m355_picture_replace.argtypes = [vp, i, vp]
        L.m355_picture_release.argtypes = [vp, i]
        L.m355_decode_resident.argtypes = [vp, i]
        L.m355_decode_batch.argtypes = [vp, ctypes.POINTER(i), i]
        L.m355_set_stages.argtypes = [vp, i]
        L.m355_set_pipeline_depth.argtypes = [vp, i]
        L.m355_timing_reset.argtypes = [vp]
        L.m355_timing_collect.argtypes = [vp, ctypes.POINTER(i), ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float)]
        L.m355_stream.argtypes = [vp]
        L.m355_stream.restype = vp
        L.m355_shard_set.argtypes = [vp, i, i]
        L.m355_shard_owner_of_tile.argtypes = [i, i, i]
        L.m355_shard_xbuf_bytes.argtypes = [vp, i, i]
        L.m355_shard_xbuf_bytes.restype = ctypes.c_int64
        L.m355_decode_phase.argtypes = [vp, i, i, vp]
        L.m355_decode_sharded.argtypes = [vp, i, i]
        L.m355_shard_set_comm.argtypes = [vp, vp]
        L.m355_rccl_unique_id.argtypes = [vp]
        L.m355_shard_rccl_init.argtypes = [vp, vp, i, i]
        L.m355_shard_rccl_selftest.argtypes = [vp, ctypes.c_size_t]
        L.m355_shard_ipc_init.argtypes = [vp, ctypes.c_char_p, i, i]
        L.m355_shard_ipc_close.argtypes = [vp]
        L.m355_group_create.argtypes = [ctypes.POINTER(vp), i, ctypes.POINTER(vp)]
        L.m355_group_destroy.argtypes = [vp]
        L.m355_group_destroy.restype = None
        L.m355_group_decode.argtypes = [vp, ctypes.POINTER(i), i]
        L.m355_shard_peers.argtypes = [vp, i, i, ctypes.POINTER(i), i]
        L.m355_shard_time_exchange.argtypes = [vp, i, i, i, ctypes.POINTER(ctypes.c_float)]
        L.init_acceleration_functions_mi355x.argtypes = [vp]
        L.m355_transform_add_batch.argtypes = [i, i, i, i, vp, ctypes.c_size_t, vp, ctypes.c_ssize_t, vp]

    def error(self):
        return (self.lib.m355_last_error() or b"").decode()

    def rccl_unique_id(self):
        """rank 0 of a tile-sharded job: the RCCL unique id (128 bytes) to hand to every rank (Context.shard_rccl_init)"""
        buf = ctypes.create_string_buffer(128)
        self.check(self.lib.m355_rccl_unique_id(buf))
        return buf.raw

    def check(self, rc):
        if rc != 0:
            raise M355Error(rc, self.error())

    def device_count(self):
        return self.lib.m355_device_count()


class Context:
    """One decoding context = one GPU, one HIP stream, a device-resident frame pool."""

    def __init__(self, lib=None, device=0):
        self.L = lib or Library()
        h = ctypes.c_void_p()
        self.L.check(self.L.lib.m355_create(device, ctypes.byref(h)))
        self.h = h
        self._geom = {}

    def close(self):
        if self.h:
            self.L.lib.m355_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- frames ----
    def frame_create(self, width, height, chroma_format_idc=1, bit_depth_luma=8, bit_depth_chroma=8):
        f = self.L.lib.m355_frame_create(self.h, width, height, chroma_format_idc, bit_depth_luma, bit_depth_chroma)
        if f < 0:
            raise M355Error(-f, self.L.error())
        self._geom[f] = (width, height, chroma_format_idc, bit_depth_luma, bit_depth_chroma)
        return f

    def frame_create_for(self, pp):
        return self.frame_create(int(pp["width"]), int(pp["height"]), int(pp["chroma_format_idc"]),
                                 int(pp["bit_depth_luma"]), int(pp["bit_depth_chroma"]))

    def frame_destroy(self, f):
        self.L.check(self.L.lib.m355_frame_destroy(self.h, f))
        self._geom.pop(f, None)

    def frame_upload(self, f, planes):
        for c, a in enumerate(planes):
            a = np.ascontiguousarray(a)
            self.L.check(self.L.lib.m355_frame_upload(self.h, f, c, a.ctypes.data, a.shape[1]))

    def frame_download(self, f):
        w, h, cf, bdl, bdc = self._geom[f]
        out = []
        for c, (pw, ph) in enumerate(worklist.plane_dims(w, h, cf)):
            if pw == 0:
                continue
            a = np.zeros((ph, pw), np.uint8 if (bdl if c == 0 else bdc) <= 8 else np.uint16)
            self.L.check(self.L.lib.m355_frame_download(self.h, f, c, a.ctypes.data, pw))
            out.append(a)
        return out

    def frame_download_async(self, f):
        """start the download of all planes behind the frame's last writer (into pinned planes); -> token for frame_download_finish"""
        w, h, cf, bdl, bdc = self._geom[f]
        dst = (ctypes.c_void_p * 3)(); strides = (ctypes.c_ssize_t * 3)()
        arrays, bufs = [], []
        for c, (pw, ph) in enumerate(worklist.plane_dims(w, h, cf)):
            if pw == 0:
                continue
            dt = np.uint8 if (bdl if c == 0 else bdc) <= 8 else np.uint16
            nbytes = pw * ph * np.dtype(dt).itemsize
            p = self.L.lib.m355_host_alloc(nbytes)
            if not p:
                for q in bufs:                       # (the planes of this call that were already allocated)
                    self.L.lib.m355_host_free(q)
                raise M355Error(4, self.L.error())   # M355_ERR_NOMEM
            bufs.append(p)
            arrays.append(np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ctypes.c_uint8)), (nbytes,)).view(dt).reshape(ph, pw))
            dst[c] = p; strides[c] = pw
        self.L.check(self.L.lib.m355_frame_download_async(self.h, f, dst, strides))
        return (f, arrays, bufs)

    def pinned_planes(self, f):
        """pinned host planes for frame f (m355_host_alloc), for repeated frame_download_start calls -> (dst[3], strides[3], arrays, pointers)"""
        w, h, cf, bdl, bdc = self._geom[f]
        dst = (ctypes.c_void_p * 3)(); strides = (ctypes.c_ssize_t * 3)()
        arrays, bufs = [], []
        for c, (pw, ph) in enumerate(worklist.plane_dims(w, h, cf)):
            if pw == 0:
                continue
            dt = np.uint8 if (bdl if c == 0 else bdc) <= 8 else np.uint16
            nbytes = pw * ph * np.dtype(dt).itemsize
            p = self.L.lib.m355_host_alloc(nbytes)
            if not p:
                for q in bufs:                       # (the planes of this call that were already allocated)
                    self.L.lib.m355_host_free(q)
                raise M355Error(4, self.L.error())   # M355_ERR_NOMEM
            bufs.append(p)
            arrays.append(np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ctypes.c_uint8)), (nbytes,)).view(dt).reshape(ph, pw))
            dst[c] = p; strides[c] = pw
        return dst, strides, arrays, bufs

    def frame_download_start(self, f, planes):
        """m355_frame_download_async into planes from pinned_planes(); frame_download_wait(f) completes it"""
        self.L.check(self.L.lib.m355_frame_download_async(self.h, f, planes[0], planes[1]))

    def frame_download_wait(self, f):
        self.L.check(self.L.lib.m355_frame_download_wait(self.h, f))

    def pinned_free(self, planes):
        for p in planes[3]:
            self.L.lib.m355_host_free(p)

    def frame_download_finish(self, token):
        f, arrays, bufs = token
        self.L.check(self.L.lib.m355_frame_download_wait(self.h, f))
        out = [a.copy() for a in arrays]
        for p in bufs:
            self.L.lib.m355_host_free(p)
        return out

    def frame_fill(self, f, luma, chroma):
        self.L.check(self.L.lib.m355_frame_fill(self.h, f, luma, chroma))

    def measure_copy_rate(self, nbytes=1 << 30, iters=9):
        """device-to-device copy rate of this box in GB/s (m355_measure_copy_rate: read + written bytes / time)"""
        g = ctypes.c_double(0)
        self.L.check(self.L.lib.m355_measure_copy_rate(self.h, nbytes, iters, ctypes.byref(g)))
        return float(g.value)

    def frame_hash(self, f, hash_type):
        """SEI decoded picture hash (m355_frame_hash): list of per-plane values — bytes (MD5) or int (CRC, checksum)"""
        out = PictureHash()
        self.L.check(self.L.lib.m355_frame_hash(self.h, f, hash_type, ctypes.addressof(out)))
        w, h, cf, bdl, bdc = self._geom[f]
        n = 3 if cf else 1
        if hash_type == HASH_MD5:
            return [bytes(out.md5[c]) for c in range(n)]
        return [int((out.crc if hash_type == HASH_CRC else out.checksum)[c]) for c in range(n)]

    # ---- pictures ----
    def submit(self, pic):
        c, keep = pic.to_c()
        self.L.check(self.L.lib.m355_submit_picture(self.h, ctypes.addressof(c)))
        del keep

    def submit_in_place(self, src_pic, slack=1.0, fill_threads=16, refill=True, state=None):
        """The in-place path: m355_arena_begin (capacities = the picture's counts x slack) -> the lists are written into the
        pinned arena by libm355synth's m355_synth_fill_arena (standing in for recorder threads) -> m355_submit_picture on
        those pointers (no host copy inside the library).  refill=False re-submits what the arena still holds from an
        earlier call with the same capacities (three arenas rotate): the library's own share of the work alone.
        `state` caches the marshalled source picture between calls."""
        from . import synth
        state = state if state is not None else {}
        if "src" not in state:
            src, state["keep"] = src_pic.to_c()
            caps = ArenaCaps()
            for n in ("n_slices", "n_ctbs", "n_cus", "n_tus", "n_pbs", "n_wts", "n_ibs"):
                setattr(caps, n, int(getattr(src, n)) if n in ("n_slices", "n_ctbs") else int(getattr(src, n) * slack) + 1)
            for b in range(4):
                caps.n_rbs[b] = int(src.rb_count[b] * slack) + 1
            caps.n_coeffs = int(src.n_coeffs * slack) + 1
            caps.n_pcm = int(src.n_pcm * slack) + 1
            caps.scaling = 1 if src.scaling_factors else 0
            state.update(src=src, caps=caps, dst=worklist.CPicture(), lib=synth._lib(), a_src=ctypes.addressof(src))
            state["a_caps"], state["a_dst"] = ctypes.addressof(caps), ctypes.addressof(state["dst"])
        src, dst, lib = state["src"], state["dst"], state["lib"]
        src.dst_frame = src_pic.dst_frame
        rc = self.L.lib.m355_arena_begin(self.h, state["a_caps"], state["a_dst"])
        if rc:
            self.L.check(rc)
        if refill or not state.get("filled", 0) >= 12:      # (every arena of the ring holds the lists)
            lib.m355_synth_fill_arena(state["a_src"], state["a_caps"], state["a_dst"], fill_threads)
            state["filled"] = state.get("filled", 0) + 1
        else:
            lib.m355_synth_fill_arena_header(state["a_src"], state["a_dst"])
        dst.dst_frame = src.dst_frame
        ctypes.memmove(dst.ref_frames, src.ref_frames, 4 * worklist.MAX_REF_FRAMES)
        rc = self.L.lib.m355_submit_picture(self.h, state["a_dst"])
        if rc:
            self.L.check(rc)
        return state

    def wait(self):
        self.L.check(self.L.lib.m355_wait(self.h))

    # ---- tile-sharded picture in one call (m355_decode_sharded) ----
    def decode_sharded(self, handle, gather=True):
        self.L.check(self.L.lib.m355_decode_sharded(self.h, handle, 1 if gather else 0))

    def shard_time_exchange(self, handle, which, iters=20):
        ms = ctypes.c_float(0)
        self.L.check(self.L.lib.m355_shard_time_exchange(self.h, handle, which, iters, ctypes.byref(ms)))
        return float(ms.value)

    def shard_set_comm(self, comm_struct):
        """comm_struct: a ctypes m355_comm (kept alive by the caller) or None"""
        self.L.check(self.L.lib.m355_shard_set_comm(self.h, ctypes.addressof(comm_struct) if comm_struct is not None else None))

    def shard_rccl_selftest(self, words=1 << 16):
        """collective: real bytes through ncclSend / ncclRecv / ncclAllGather of this context's communicator, checked on the host"""
        self.L.check(self.L.lib.m355_shard_rccl_selftest(self.h, words))

    def shard_rccl_init(self, unique_id, rank, nranks):
        buf = ctypes.create_string_buffer(bytes(unique_id), 128)
        self.L.check(self.L.lib.m355_shard_rccl_init(self.h, buf, rank, nranks))

    def shard_ipc_init(self, name, rank, nranks):
        """tile sharding across the rank processes of one node without a collective library (m355_shard_ipc_init): `name` = a job-unique string, the same on every rank"""
        self.L.check(self.L.lib.m355_shard_ipc_init(self.h, name.encode(), rank, nranks))

    def shard_ipc_close(self):
        self.L.check(self.L.lib.m355_shard_ipc_close(self.h))

    def last_serial(self):
        """serial of the decode the last submit / decode call enqueued"""
        return int(self.L.lib.m355_last_serial(self.h))

    def decode_status(self, serial):
        """non-blocking: 0 finished and fine, 6 (M355_ERR_BUSY) still running, 3 (M355_ERR_INVALID) lists rejected on the device"""
        return int(self.L.lib.m355_decode_status(self.h, serial))

    def upload(self, pic):
        c, keep = pic.to_c()
        r = self.L.lib.m355_picture_upload(self.h, ctypes.addressof(c))
        del keep
        if r < 0:
            raise M355Error(-r, self.L.error())
        return r

    def upload_in_place(self, src_pic, handle=-1, slack=1.0, fill_threads=4):
        """The in-place path of a RESIDENT picture (m355_picture_arena_begin -> the lists are written into the handle's pinned arena
        by libm355synth, standing in for recorder threads -> m355_picture_replace copies nothing on the host); what a tile-sharded
        context offers instead of m355_arena_begin.  -> handle"""
        from . import synth
        src, keep = src_pic.to_c()
        caps = ArenaCaps()
        for n in ("n_slices", "n_ctbs", "n_cus", "n_tus", "n_pbs", "n_wts", "n_ibs"):
            setattr(caps, n, int(getattr(src, n)) if n in ("n_slices", "n_ctbs") else int(getattr(src, n) * slack) + 1)
        for b in range(4):
            caps.n_rbs[b] = int(src.rb_count[b] * slack) + 1
        caps.n_coeffs = int(src.n_coeffs * slack) + 1
        caps.n_pcm = int(src.n_pcm * slack) + 1
        caps.scaling = 1 if src.scaling_factors else 0
        dst = worklist.CPicture()
        h = self.L.lib.m355_picture_arena_begin(self.h, handle, ctypes.addressof(caps), ctypes.addressof(src) + worklist.CPicture.pp.offset, ctypes.addressof(dst))
        if h < 0:
            raise M355Error(-h, self.L.error())
        synth._lib().m355_synth_fill_arena(ctypes.addressof(src), ctypes.addressof(caps), ctypes.addressof(dst), fill_threads)
        dst.dst_frame = src.dst_frame
        ctypes.memmove(dst.ref_frames, src.ref_frames, 4 * worklist.MAX_REF_FRAMES)
        self.L.check(self.L.lib.m355_picture_replace(self.h, h, ctypes.addressof(dst)))
        del keep
        return h

    def release(self, handle):
        self.L.check(self.L.lib.m355_picture_release(self.h, handle))

    def decode_resident(self, handle):
        self.L.check(self.L.lib.m355_decode_resident(self.h, handle))

    def decode_batch(self, handles):
        """several independent intra pictures with one intra stage (m355_decode_batch)"""
        a = (ctypes.c_int * len(handles))(*handles)
        self.L.check(self.L.lib.m355_decode_batch(self.h, a, len(handles)))

    def set_stages(self, mask):
        self.L.check(self.L.lib.m355_set_stages(self.h, mask))

    def set_pipeline_depth(self, depth):
        self.L.check(self.L.lib.m355_set_pipeline_depth(self.h, depth))

    def timing_reset(self):
        self.L.check(self.L.lib.m355_timing_reset(self.h))

    def timing_collect(self):
        """-> (n_decodes, avg_total_ms, {stage: avg_ms}) for all decodes since timing_reset()"""
        n = ctypes.c_int()
        total = ctypes.c_float()
        st = (ctypes.c_float * 6)()
        self.L.check(self.L.lib.m355_timing_collect(self.h, ctypes.byref(n), ctypes.byref(total), st))
        return n.value, total.value, dict(zip(["meta", "inter", "residual", "intra", "deblock", "sao"], list(st)))

    def stream(self):
        return self.L.lib.m355_stream(self.h)

    # ---- tile sharding (see libde265_amd/shard.py for the multi-GPU driver) ----
    def shard_set(self, rank, nranks):
        self.L.check(self.L.lib.m355_shard_set(self.h, rank, nranks))

    def shard_xbuf_bytes(self, handle, which):
        n = self.L.lib.m355_shard_xbuf_bytes(self.h, handle, which)
        if n < 0:
            raise M355Error(-n, self.L.error())
        return n

    def decode_phase(self, handle, phase, xbuf_ptr):
        self.L.check(self.L.lib.m355_decode_phase(self.h, handle, phase, xbuf_ptr))


# ---------------------------------------------------------------------------------------------------
# Slot layer: ctypes mirror of `struct m355_acceleration_functions` (= the reference's
# `struct acceleration_functions`, libde265/acceleration.h:29-231), member for member.
# ---------------------------------------------------------------------------------------------------
_vp, _i, _sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_ssize_t
_F = lambda *a: ctypes.CFUNCTYPE(None, *a)  # noqa: E731  (all slots return void)
_WAVG8, _UNW8 = _F(_vp, _sz, _vp, _vp, _sz, _i, _i), _F(_vp, _sz, _vp, _sz, _i, _i)
_W8, _WB8 = _F(_vp, _sz, _vp, _sz, _i, _i, _i, _i, _i), _F(_vp, _sz, _vp, _vp, _sz, _i, _i, _i, _i, _i, _i, _i)
_WAVG16, _UNW16 = _F(_vp, _sz, _vp, _vp, _sz, _i, _i, _i), _F(_vp, _sz, _vp, _sz, _i, _i, _i)
_W16, _WB16 = _F(_vp, _sz, _vp, _sz, _i, _i, _i, _i, _i, _i), _F(_vp, _sz, _vp, _vp, _sz, _i, _i, _i, _i, _i, _i, _i, _i)
_EPEL8 = _F(_vp, _sz, _vp, _sz, _i, _i, _i, _i, _vp)
_EPEL = _F(_vp, _sz, _vp, _sz, _i, _i, _i, _i, _vp, _i)
_QPEL8, _QPEL16 = _F(_vp, _sz, _vp, _sz, _i, _i, _vp), _F(_vp, _sz, _vp, _sz, _i, _i, _vp, _i)
_BYP, _TS8, _TSR8, _TA8 = _F(_vp, _vp, _i), _F(_vp, _vp, _sz), _F(_vp, _vp, _i, _sz), _F(_vp, _vp, _sz)
_TA16, _ROT, _IDCT = _F(_vp, _vp, _sz, _i), _F(_vp, _i), _F(_vp, _vp, _i, _i)
_ADDR, _DEQ = _F(_vp, _sz, _vp, _i, _i), _F(_vp, _vp, _vp, _i, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32)
_DBL, _DBC, _RDP = _F(_vp, _sz, _i, _i, _i, _i, _i, _i, _i), _F(_vp, _sz, _i, _i, _i, _i), _F(_vp, _vp, _i, _i, _i)
_IP, _IA = _F(_vp, _sz, _i, _i, _vp), _F(_vp, _sz, _i, _i, _i, _i, _i, _i, _i, _vp)
_FWD = _F(_vp, _vp, _sz)


class AccelerationFunctions(ctypes.Structure):
    _fields_ = [
        ("put_weighted_pred_avg_8", _WAVG8), ("put_unweighted_pred_8", _UNW8), ("put_weighted_pred_8", _W8),
        ("put_weighted_bipred_8", _WB8),
        ("put_weighted_pred_avg_16", _WAVG16), ("put_unweighted_pred_16", _UNW16), ("put_weighted_pred_16", _W16),
        ("put_weighted_bipred_16", _WB16),
        ("put_hevc_epel_8", _EPEL8), ("put_hevc_epel_h_8", _EPEL), ("put_hevc_epel_v_8", _EPEL), ("put_hevc_epel_hv_8", _EPEL),
        ("put_hevc_qpel_8", (_QPEL8 * 4) * 4),
        ("put_hevc_epel_16", _EPEL), ("put_hevc_epel_h_16", _EPEL), ("put_hevc_epel_v_16", _EPEL), ("put_hevc_epel_hv_16", _EPEL),
        ("put_hevc_qpel_16", (_QPEL16 * 4) * 4),
        ("transform_bypass", _BYP), ("transform_bypass_rdpcm_v", _BYP), ("transform_bypass_rdpcm_h", _BYP),
        ("transform_skip_8", _TS8), ("transform_skip_rdpcm_v_8", _TSR8), ("transform_skip_rdpcm_h_8", _TSR8),
        ("transform_4x4_dst_add_8", _TA8), ("transform_add_8", _TA8 * 4),
        ("transform_skip_16", _TA16), ("transform_4x4_dst_add_16", _TA16), ("transform_add_16", _TA16 * 4),
        ("rotate_coefficients", _ROT), ("transform_idst_4x4", _IDCT), ("transform_idct_4x4", _IDCT),
        ("transform_idct_8x8", _IDCT), ("transform_idct_16x16", _IDCT), ("transform_idct_32x32", _IDCT),
        ("add_residual_8", _ADDR), ("add_residual_16", _ADDR),
        ("dequant_coeff_block", _DEQ),
        ("deblock_luma_8", _DBL), ("deblock_chroma_8", _DBC),
        ("rdpcm_v", _RDP), ("rdpcm_h", _RDP), ("transform_skip_residual", _RDP),
        ("intra_pred_dc_8", _IP), ("intra_pred_dc_16", _IP), ("intra_pred_planar_8", _IP), ("intra_pred_planar_16", _IP),
        ("intra_pred_angular_8", _IA), ("intra_pred_angular_16", _IA),
        ("fwd_transform_4x4_dst_8", _FWD), ("fwd_transform_8", _FWD * 4), ("hadamard_transform_8", _FWD * 4),
    ]


assert ctypes.sizeof(AccelerationFunctions) == 94 * ctypes.sizeof(ctypes.c_void_p)


def acceleration_functions(lib=None):
    """A table filled by init_acceleration_functions_mi355x() (raises when no device is visible)."""
    lib = lib or Library()
    t = AccelerationFunctions()
    rc = lib.lib.init_acceleration_functions_mi355x(ctypes.byref(t))
    if rc != 0:
        raise M355Error(rc, "init_acceleration_functions_mi355x: no HIP device (there is no CPU fallback)")
    return t


class Group:
    """Tile sharding inside one process (m355_group_*): rank r = ctxs[r]; each context uploads its share of the picture
    (shard.shard_picture) and decode() issues every rank's phases with the exchanges as copies between the contexts."""

    def __init__(self, lib, ctxs):
        self.L, self.ctxs = lib, list(ctxs)
        arr = (ctypes.c_void_p * len(self.ctxs))(*[c.h for c in self.ctxs])
        self.h = ctypes.c_void_p()
        lib.check(lib.lib.m355_group_create(arr, len(self.ctxs), ctypes.byref(self.h)))

    def decode(self, handles, gather=True):
        arr = (ctypes.c_int * len(handles))(*handles)
        self.L.check(self.L.lib.m355_group_decode(self.h, arr, 1 if gather else 0))

    def wait(self):
        for c in self.ctxs:
            c.wait()

    def close(self):
        if self.h:
            self.L.lib.m355_group_destroy(self.h)
            self.h = None

