"""Work-list structures of the picture layer — numpy mirrors of the PODs in include/de265_mi355x.h.

A `Picture` holds one picture's lists as numpy structured arrays; `to_c()` produces the ctypes
`m355_picture` the C ABI (and the oracle) consume; `dumps()/loads()` (de)serialise the portable
"M355WL01" blob used for golden fixtures (tests/golden/) and by the synthetic generator.
"""
import ctypes
import struct

import numpy as np

MAX_TILE_COLS = 20
MAX_TILE_ROWS = 22
MAX_REF_FRAMES = 32

# picture flags (M355_PF_*)
PF_CONSTRAINED_INTRA_PRED = 1 << 0
PF_STRONG_INTRA_SMOOTHING = 1 << 1
PF_PCM_LOOP_FILTER_DISABLE = 1 << 2
PF_LF_ACROSS_TILES = 1 << 3
PF_SAO_ENABLED = 1 << 4
PF_INTRA_SMOOTHING_DISABLED = 1 << 5
PF_IMPLICIT_RDPCM = 1 << 6
PF_SCALING_LIST = 1 << 7
PF_DEBLOCK_ENABLED = 1 << 8
PF_CROSS_COMPONENT_PRED = 1 << 9
PF_TRANSFORM_SKIP_ROTATION = 1 << 10
PF_CLEAR_DST = 1 << 11
# slice flags
SF_DEBLOCK_DISABLED, SF_LF_ACROSS_SLICES, SF_SAO_LUMA, SF_SAO_CHROMA = 1, 2, 4, 8
CTBF_HAS_PCM_OR_BYPASS = 1
CUF_PCM, CUF_TRANSQUANT_BYPASS = 1, 2
TUF_NONZERO_COEFF = 1
PBF_PRED_L0, PBF_PRED_L1, PBF_MC_L0, PBF_MC_L1, PBF_WEIGHTED, PBF_FILL_L0, PBF_FILL_L1 = 1, 2, 4, 8, 16, 32, 64
RK_DCT, RK_DST, RK_SKIP, RK_BYPASS = 0, 1, 2, 3
RBF_DEFERRED, RBF_RDPCM_H, RBF_RDPCM_V, RBF_ROTATE, RBF_DEQUANTIZED = 1, 2, 4, 8, 16
IBF_HAS_RESIDUAL, IBF_DISABLE_BOUNDARY_FILTER, IBF_PCM = 1, 2, 4
STAGE_INTER, STAGE_RESIDUAL, STAGE_INTRA, STAGE_DEBLOCK, STAGE_SAO, STAGE_ALL = 1, 2, 4, 8, 16, 31

PIC_PARAMS = np.dtype([
    ("width", "<i4"), ("height", "<i4"),
    ("chroma_format_idc", "u1"), ("bit_depth_luma", "u1"), ("bit_depth_chroma", "u1"),
    ("log2_ctb_size", "u1"), ("log2_min_tb_size", "u1"), ("log2_min_cb_size", "u1"),
    ("pic_cb_qp_offset", "i1"), ("pic_cr_qp_offset", "i1"),
    ("flags", "<u4"),
    ("num_tile_cols", "u1"), ("num_tile_rows", "u1"),
    ("col_bd", "<u2", (MAX_TILE_COLS + 1,)), ("row_bd", "<u2", (MAX_TILE_ROWS + 1,)),
    ("reserved", "<u2"),
])
SLICE = np.dtype([("slice_addr_rs", "<i4"), ("beta_offset", "i1"), ("tc_offset", "i1"), ("flags", "u1"), ("reserved", "u1")])
CTB = np.dtype([("slice_idx", "<u2"), ("sao_type", "u1"), ("sao_eo_class", "u1"), ("sao_band_pos", "u1", (3,)),
                ("flags", "u1"), ("sao_offset", "i1", (3, 4)), ("ib_start", "<u4"), ("ib_count", "<u4")])
CU = np.dtype([("x", "<u2"), ("y", "<u2"), ("log2_size", "u1"), ("pred_mode", "u1"), ("part_mode", "u1"),
               ("qp_y", "i1"), ("flags", "u1"), ("reserved", "u1", (3,))])
TU = np.dtype([("x", "<u2"), ("y", "<u2"), ("log2_size", "u1"), ("flags", "u1"), ("reserved", "<u2")])
WT = np.dtype([("w", "<i2", (3,)), ("o", "<i2", (3,)), ("log2wd_luma", "u1"), ("log2wd_chroma", "u1"), ("reserved", "<u2")])
PB = np.dtype([("x", "<u2"), ("y", "<u2"), ("w", "u1"), ("h", "u1"), ("flags", "u1"), ("reserved", "u1"),
               ("ref_slot", "i1", (2,)), ("mv", "<i2", (2, 2)), ("wt_idx", "<u2", (2,)), ("reserved2", "<u2")])
RB = np.dtype([("x", "<u2"), ("y", "<u2"), ("cidx", "u1"), ("log2_size", "u1"), ("kind", "u1"), ("flags", "u1"),
               ("qp", "u1"), ("matrix_id", "u1"), ("ncoeff", "<u2"), ("coeff_ofs", "<u4"), ("res_ofs", "<u4")])
IB = np.dtype([("x", "<u2"), ("y", "<u2"), ("cidx", "u1"), ("log2_size", "u1"), ("mode", "u1"), ("flags", "u1"),
               ("res_ofs", "<u4")])

assert PIC_PARAMS.itemsize == 112 and SLICE.itemsize == 8 and CTB.itemsize == 28 and CU.itemsize == 12
assert TU.itemsize == 8 and WT.itemsize == 16 and PB.itemsize == 24 and RB.itemsize == 20 and IB.itemsize == 12

SCALING_BYTES = 6 * (16 + 64 + 256 + 1024)


class CPicParams(ctypes.Structure):
    _fields_ = [("raw", ctypes.c_uint8 * 112)]


class CPicture(ctypes.Structure):
    """ctypes mirror of `m355_picture`."""
    _fields_ = [
        ("pp", CPicParams),
        ("dst_frame", ctypes.c_int32),
        ("ref_frames", ctypes.c_int32 * MAX_REF_FRAMES),
        ("n_slices", ctypes.c_int32), ("n_ctbs", ctypes.c_int32), ("n_cus", ctypes.c_int32),
        ("n_tus", ctypes.c_int32), ("n_pbs", ctypes.c_int32), ("n_wts", ctypes.c_int32), ("n_ibs", ctypes.c_int32),
        ("rb_count", ctypes.c_int32 * 4),
        ("n_coeffs", ctypes.c_uint32), ("n_pcm", ctypes.c_uint32), ("res_len", ctypes.c_uint32),
        ("slices", ctypes.c_void_p), ("ctbs", ctypes.c_void_p), ("cus", ctypes.c_void_p), ("tus", ctypes.c_void_p),
        ("pbs", ctypes.c_void_p), ("wts", ctypes.c_void_p), ("rbs", ctypes.c_void_p), ("ibs", ctypes.c_void_p),
        ("coeffs", ctypes.c_void_p), ("pcm", ctypes.c_void_p), ("scaling_factors", ctypes.c_void_p),
    ]


_LISTS = [("slices", SLICE), ("ctbs", CTB), ("cus", CU), ("tus", TU), ("pbs", PB), ("wts", WT), ("rbs", RB),
          ("ibs", IB), ("coeffs", np.dtype("<u4")), ("pcm", np.dtype("<u2"))]
MAGIC = b"M355WL01"


class Picture:
    """One picture's work lists (host side)."""

    def __init__(self):
        self.pp = np.zeros(1, PIC_PARAMS)
        self.dst_frame = 0
        self.ref_frames = [-1] * MAX_REF_FRAMES
        for name, dt in _LISTS:
            setattr(self, name, np.zeros(0, dt))
        self.rb_count = [0, 0, 0, 0]
        self.res_len = 0
        self.scaling_factors = None
        self.meta = {}          # free-form (poc, expected md5 ...), not part of the ABI

    # ---- geometry helpers ----
    @property
    def ctb_size(self):
        return 1 << int(self.pp["log2_ctb_size"][0])

    @property
    def pic_w_ctbs(self):
        return (int(self.pp["width"][0]) + self.ctb_size - 1) // self.ctb_size

    @property
    def pic_h_ctbs(self):
        return (int(self.pp["height"][0]) + self.ctb_size - 1) // self.ctb_size

    def set_single_tile(self):
        self.pp["num_tile_cols"] = 1
        self.pp["num_tile_rows"] = 1
        self.pp["col_bd"][0, :2] = [0, self.pic_w_ctbs]
        self.pp["row_bd"][0, :2] = [0, self.pic_h_ctbs]

    # ---- C view ----
    def to_c(self):
        """Returns (CPicture, keepalive). Arrays are made contiguous; keepalive must outlive the call."""
        c = CPicture()
        keep = []
        raw = np.ascontiguousarray(self.pp).view(np.uint8)
        ctypes.memmove(c.pp.raw, raw.ctypes.data, 112)
        c.dst_frame = int(self.dst_frame)
        for i in range(MAX_REF_FRAMES):
            c.ref_frames[i] = int(self.ref_frames[i])
        for name, dt in _LISTS:
            a = np.ascontiguousarray(getattr(self, name), dtype=dt)
            keep.append(a)
            setattr(c, name, a.ctypes.data if a.size else None)
        c.n_slices, c.n_ctbs, c.n_cus, c.n_tus = len(self.slices), len(self.ctbs), len(self.cus), len(self.tus)
        c.n_pbs, c.n_wts, c.n_ibs = len(self.pbs), len(self.wts), len(self.ibs)
        assert sum(self.rb_count) == len(self.rbs)
        for i in range(4):
            c.rb_count[i] = int(self.rb_count[i])
        c.n_coeffs, c.n_pcm, c.res_len = len(self.coeffs), len(self.pcm), int(self.res_len)
        if self.scaling_factors is not None:
            sf = np.ascontiguousarray(self.scaling_factors, dtype=np.uint8)
            assert sf.size == SCALING_BYTES
            keep.append(sf)
            c.scaling_factors = sf.ctypes.data
        else:
            c.scaling_factors = None
        return c, keep

    # ---- portable blob ----
    def dumps(self):
        out = [MAGIC, np.ascontiguousarray(self.pp).tobytes()]
        out.append(struct.pack("<i", int(self.dst_frame)))
        out.append(struct.pack("<%di" % MAX_REF_FRAMES, *[int(x) for x in self.ref_frames]))
        counts = [len(self.slices), len(self.ctbs), len(self.cus), len(self.tus), len(self.pbs), len(self.wts),
                  len(self.ibs)] + [int(x) for x in self.rb_count] + [len(self.coeffs), len(self.pcm), int(self.res_len),
                                                                       1 if self.scaling_factors is not None else 0]
        out.append(struct.pack("<15i", *counts))
        for name, dt in _LISTS:
            b = np.ascontiguousarray(getattr(self, name), dtype=dt).tobytes()
            out.append(b + b"\0" * (-len(b) % 4))
        if self.scaling_factors is not None:
            out.append(np.ascontiguousarray(self.scaling_factors, dtype=np.uint8).tobytes())
        return b"".join(out)

    @staticmethod
    def loads(buf, offset=0):
        """Parse one blob at `offset`; returns (Picture, next_offset)."""
        assert buf[offset:offset + 8] == MAGIC, "bad work-list magic"
        p = Picture()
        o = offset + 8
        p.pp = np.frombuffer(buf, PIC_PARAMS, 1, o).copy(); o += 112
        p.dst_frame = struct.unpack_from("<i", buf, o)[0]; o += 4
        p.ref_frames = list(struct.unpack_from("<%di" % MAX_REF_FRAMES, buf, o)); o += 4 * MAX_REF_FRAMES
        c = struct.unpack_from("<15i", buf, o); o += 60
        n = {"slices": c[0], "ctbs": c[1], "cus": c[2], "tus": c[3], "pbs": c[4], "wts": c[5], "ibs": c[6],
             "rbs": sum(c[7:11]), "coeffs": c[11], "pcm": c[12]}
        p.rb_count = list(c[7:11]); p.res_len = c[13]
        for name, dt in _LISTS:
            cnt = n[name]
            setattr(p, name, np.frombuffer(buf, dt, cnt, o).copy())
            nb = cnt * dt.itemsize
            o += nb + (-nb % 4)
        if c[14]:
            p.scaling_factors = np.frombuffer(buf, np.uint8, SCALING_BYTES, o).copy(); o += SCALING_BYTES
        return p, o


def plane_dims(width, height, chroma_format_idc):
    """(w,h) of the three planes (image.cc:113-117 / sps SubWidthC,SubHeightC)."""
    sw = 2 if chroma_format_idc in (1, 2) else 1
    sh = 2 if chroma_format_idc == 1 else 1
    if chroma_format_idc == 0:
        return [(width, height), (0, 0), (0, 0)]
    return [(width, height), (width // sw, height // sh), (width // sw, height // sh)]
