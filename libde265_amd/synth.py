"""Synthetic work lists for the benchmark configs (BASELINE.json / SURVEY.md §8d) — Python face of
csrc/synth.c (libm355synth.so, plain host C; no GPU involved)."""
import ctypes
import os

import numpy as np

from . import worklist

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class SynthCfg(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in
                ("width", "height", "bit_depth", "log2_ctb", "tile_cols", "tile_rows", "intra_pct", "bipred_pct",
                 "weighted_pct", "oob_mv_pct", "cbf_pct", "deblock", "sao", "n_refs", "lf_across_tiles")] + \
               [("seed", ctypes.c_uint32), ("fixed_cu_log2", ctypes.c_int32), ("n_slices", ctypes.c_int32),
                ("features", ctypes.c_int32), ("chroma_format", ctypes.c_int32)]

SYN_CONSTRAINED_INTRA, SYN_TRANSQUANT_BYPASS, SYN_SCALING_LIST, SYN_PCM, SYN_PCM_LOOP_FILTER_DISABLE, SYN_CROSS_COMPONENT = 1, 2, 4, 8, 16, 32   # SynthCfg.features bits (csrc/synth.c)
SYN_RDPCM, SYN_ROTATE, SYN_MISSING_REF, SYN_DEQUANTIZED = 64, 128, 256, 512


class _SynthOut(ctypes.Structure):
    _fields_ = [("pic", worklist.CPicture), ("owned", ctypes.c_void_p * 16)]


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libm355synth.so")
        if not os.path.exists(path):
            raise RuntimeError("libm355synth.so missing: make -C libde265_amd/csrc")
        _LIB = ctypes.CDLL(path)
        _LIB.m355_synth_picture.argtypes = [ctypes.POINTER(SynthCfg), ctypes.POINTER(_SynthOut)]
        _LIB.m355_synth_free.argtypes = [ctypes.POINTER(_SynthOut)]
        _LIB.m355_synth_ref_plane.argtypes = [ctypes.c_uint32, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        _LIB.m355_synth_fill_arena.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        _LIB.m355_synth_fill_arena.restype = None
        _LIB.m355_synth_fill_arena_header.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        _LIB.m355_synth_fill_arena_header.restype = None
    return _LIB


# the configs of BASELINE.json (C2..C5); seeds as suggested in SURVEY.md §8d
CONFIGS = {
    "c2_1080p_intra": dict(width=1920, height=1080, bit_depth=8, tile_cols=1, tile_rows=1, intra_pct=100, n_refs=0,
                           deblock=0, sao=1, seed=0xC2C2C2C2),
    "c3_4k_inter": dict(width=3840, height=2160, bit_depth=8, tile_cols=1, tile_rows=1, intra_pct=3, n_refs=2,
                        deblock=1, sao=0, seed=0xC3C3C3C3),
    "c4_4k_4tiles": dict(width=3840, height=2160, bit_depth=8, tile_cols=2, tile_rows=2, intra_pct=3, n_refs=2,
                         deblock=1, sao=1, seed=0xC4C4C4C4),
    "c5_8k10_8tiles": dict(width=7680, height=4320, bit_depth=10, tile_cols=4, tile_rows=2, intra_pct=3, n_refs=2,
                           deblock=1, sao=1, seed=0xC5C5C5C5),
    # plumbing checks of bench.py itself on the CPU tier (SIMT-interpreter library; tests/test_bench_launch.py): never a benchmark
    "tiny_4tiles": dict(width=256, height=192, bit_depth=8, tile_cols=2, tile_rows=2, intra_pct=5, n_refs=2, deblock=1, sao=1, seed=0x71117111),
    # diagnostics (not BASELINE configs): C5 with one CU size only / without out-of-picture motion vectors — what the block mix costs
    "c5x_cu64": dict(width=7680, height=4320, bit_depth=10, tile_cols=4, tile_rows=2, intra_pct=3, n_refs=2, deblock=1, sao=1,
                     seed=0xC5C5C5C5, fixed_cu_log2=6, oob_mv_pct=0),
    "c5x_cu16": dict(width=7680, height=4320, bit_depth=10, tile_cols=4, tile_rows=2, intra_pct=3, n_refs=2, deblock=1, sao=1,
                     seed=0xC5C5C5C5, fixed_cu_log2=4, oob_mv_pct=0),
    # ... without explicit weights and out-of-picture vectors (k_inter_jobs' two main classes only), all PBs predicted from one list / from two
    "c5x_plain": dict(width=7680, height=4320, bit_depth=10, tile_cols=4, tile_rows=2, intra_pct=3, n_refs=2, deblock=1, sao=1,
                      seed=0xC5C5C5C5, oob_mv_pct=0, weighted_pct=0),
    "c5x_uni": dict(width=7680, height=4320, bit_depth=10, tile_cols=4, tile_rows=2, intra_pct=3, n_refs=2, deblock=1, sao=1,
                    seed=0xC5C5C5C5, oob_mv_pct=0, weighted_pct=0, bipred_pct=0),
    "c5x_bi": dict(width=7680, height=4320, bit_depth=10, tile_cols=4, tile_rows=2, intra_pct=3, n_refs=2, deblock=1, sao=1,
                   seed=0xC5C5C5C5, oob_mv_pct=0, weighted_pct=0, bipred_pct=100),
    "c5x_noedge": dict(width=7680, height=4320, bit_depth=10, tile_cols=4, tile_rows=2, intra_pct=3, n_refs=2, deblock=1, sao=1,
                       seed=0xC5C5C5C5, oob_mv_pct=0),
}


def make_cfg(**kw):
    d = dict(width=416, height=240, bit_depth=8, log2_ctb=6, tile_cols=1, tile_rows=1, intra_pct=10, bipred_pct=50,
             weighted_pct=10, oob_mv_pct=2, cbf_pct=60, deblock=1, sao=1, n_refs=2, lf_across_tiles=1, seed=1,
             fixed_cu_log2=0, n_slices=0, features=0, chroma_format=0)
    d.update(kw)
    c = SynthCfg()
    for k, v in d.items():
        setattr(c, k, v)
    return c


def _copy(ptr, n, dt):
    if not ptr or n == 0:
        return np.zeros(0, dt)
    buf = (ctypes.c_uint8 * (n * dt.itemsize)).from_address(ptr)
    return np.frombuffer(buf, dt, n).copy()


def picture(**kw):
    """Generate one picture's work lists -> worklist.Picture (dst_frame / ref_frames left unset)."""
    cfg = make_cfg(**kw)
    out = _SynthOut()
    rc = _lib().m355_synth_picture(ctypes.byref(cfg), ctypes.byref(out))
    if rc != 0:
        raise ValueError("m355_synth_picture failed: %d" % rc)
    try:
        c = out.pic
        p = worklist.Picture()
        p.pp = np.frombuffer(bytes(bytearray(c.pp.raw)), worklist.PIC_PARAMS, 1).copy()
        p.slices = _copy(c.slices, c.n_slices, worklist.SLICE)
        p.ctbs = _copy(c.ctbs, c.n_ctbs, worklist.CTB)
        p.cus = _copy(c.cus, c.n_cus, worklist.CU)
        p.tus = _copy(c.tus, c.n_tus, worklist.TU)
        p.pbs = _copy(c.pbs, c.n_pbs, worklist.PB)
        p.wts = _copy(c.wts, c.n_wts, worklist.WT)
        p.rb_count = [c.rb_count[i] for i in range(4)]
        p.rbs = _copy(c.rbs, sum(p.rb_count), worklist.RB)
        p.ibs = _copy(c.ibs, c.n_ibs, worklist.IB)
        p.coeffs = _copy(c.coeffs, c.n_coeffs, np.dtype("<u4"))
        p.res_len = c.res_len
        p.pcm = _copy(c.pcm, c.n_pcm, np.dtype("<u2"))
        if c.scaling_factors:
            p.scaling_factors = _copy(c.scaling_factors, worklist.SCALING_BYTES, np.dtype("u1"))
        p.meta = {"cfg": dict((f[0], getattr(cfg, f[0])) for f in SynthCfg._fields_ if f[0] != "reserved")}
    finally:
        _lib().m355_synth_free(ctypes.byref(out))
    return p


def ref_planes(seed, width, height, chroma_format_idc, bit_depth):
    out = []
    for c, (w, h) in enumerate(worklist.plane_dims(width, height, chroma_format_idc)):
        if w == 0:
            continue
        a = np.zeros((h, w), np.uint8 if bit_depth <= 8 else np.uint16)
        _lib().m355_synth_ref_plane((seed * 3 + c + 1) & 0xFFFFFFFF, w, h, bit_depth, a.ctypes.data)
        out.append(a)
    return out


def algorithmic_bytes(p):
    """Algorithmic HBM bytes of one picture, per stage, by the accounting of SURVEY.md §8(d):
    recon write S*B; coefficients 4*nnz + 16 B per coded TB; MC read per PB and list
    [(w+7)(h+7)+2(w/2+3)(h/2+3)]*B; deblock 2*S*B + 320 B per CTB; SAO 2*S*B + 17 B per CTB."""
    pp = p.pp[0]
    B = 1 if pp["bit_depth_luma"] <= 8 else 2
    cf = int(pp["chroma_format_idc"])
    S = sum(w * h for (w, h) in worklist.plane_dims(int(pp["width"]), int(pp["height"]), cf))
    nctb = len(p.ctbs)
    mc = 0
    if len(p.pbs):
        w = p.pbs["w"].astype(np.int64); h = p.pbs["h"].astype(np.int64)
        nl = ((p.pbs["flags"] & worklist.PBF_MC_L0) != 0).astype(np.int64) + ((p.pbs["flags"] & worklist.PBF_MC_L1) != 0)
        per = (w + 7) * (h + 7) + (2 * (w // 2 + 3) * (h // 2 + 3) if cf == 1 else 0)
        mc = int((per * nl).sum()) * B
        inter_samples = int((w * h).sum()) * 3 // 2 if cf == 1 else int((w * h).sum())
    else:
        inter_samples = 0
    nrb = int(sum(p.rb_count))
    coeff = 4 * len(p.coeffs) + 16 * nrb
    res_samples = int((1 << (2 * p.rbs["log2_size"].astype(np.int64))).sum()) if nrb else 0
    intra_samples = int((1 << (2 * p.ibs["log2_size"].astype(np.int64))).sum()) if len(p.ibs) else 0
    out = {
        "inter": mc + inter_samples * B,
        "residual": coeff + res_samples * B,      # coefficient read; the recon write is accounted once below
        "intra": intra_samples * B,
        "deblock": (2 * S * B + 320 * nctb) if (pp["flags"] & worklist.PF_DEBLOCK_ENABLED) else 0,
        "sao": (2 * S * B + 17 * nctb) if (pp["flags"] & worklist.PF_SAO_ENABLED) else 0,
    }
    # SURVEY's per-picture total: recon write S*B + coefficients + MC read + deblock + SAO
    out["total"] = S * B + coeff + mc + out["deblock"] + out["sao"]
    out["samples"] = S
    return out
