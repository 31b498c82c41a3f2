"""Slot-layer parity body shared by the CPU tier (kernels under the SIMT interpreter) and the GPU tier:
drive a table filled by init_acceleration_functions_mi355x() exactly like the reference's
dev-tools/test-*.cc drive its SIMD tables — identical xorshift32 inputs to our slot and to the oracle
(itself pinned to the reference's fallback-* functions in test_oracle_vs_ref.py), byte-identical
WHOLE strided buffers required (so a write outside the block is caught too)."""
from ctypes import c_ssize_t

import numpy as np

from test_oracle_vs_ref import coeff_scenarios
from util import XorShift32, pixel_dtype, ptr, ptr_at


def _bds(quick):
    return [8, 10] if quick else [8, 9, 10, 12]


def check_weighted(tab, oracle, quick):
    for bd in _bds(quick):
        rng = XorShift32(0x51071000 + bd)
        pd, pb = pixel_dtype(bd), (1 if bd <= 8 else 2)
        sfx = "_8" if pb == 1 else "_16"
        extra = [] if pb == 1 else [bd]
        for (w, h) in ([(4, 4), (16, 12)] if quick else [(2, 2), (4, 4), (8, 4), (16, 16), (12, 16), (64, 64), (6, 8)]):
            ss, ds = 64, w + 5
            s1 = rng.array(ss * h, -(1 << 13), (1 << 14) - 1, np.int16)
            s2 = rng.array(ss * h, -(1 << 13), (1 << 14) - 1, np.int16)
            base = rng.array(ds * h, 0, (1 << bd) - 1, pd)
            shift1 = max(2, 14 - bd)
            wt1, wt2, o1, o2, log2WD = rng.range(-128, 127), rng.range(-128, 127), rng.range(-128, 127) << (bd - 8), rng.range(-128, 127) << (bd - 8), rng.range(0, 7) + shift1
            a = base.copy(); b = base.copy()
            getattr(tab, "put_unweighted_pred" + sfx)(ptr(a), ds, ptr(s1), ss, w, h, *extra)
            oracle.o_put_unweighted_pred(ptr(b), c_ssize_t(ds), pb, ptr(s1), c_ssize_t(ss), w, h, bd)
            assert np.array_equal(a, b), ("unweighted", bd, w, h)
            a = base.copy(); b = base.copy()
            getattr(tab, "put_weighted_pred_avg" + sfx)(ptr(a), ds, ptr(s1), ptr(s2), ss, w, h, *extra)
            oracle.o_put_weighted_pred_avg(ptr(b), c_ssize_t(ds), pb, ptr(s1), ptr(s2), c_ssize_t(ss), w, h, bd)
            assert np.array_equal(a, b), ("avg", bd, w, h)
            a = base.copy(); b = base.copy()
            getattr(tab, "put_weighted_pred" + sfx)(ptr(a), ds, ptr(s1), ss, w, h, wt1, o1, log2WD, *extra)
            oracle.o_put_weighted_pred(ptr(b), c_ssize_t(ds), pb, ptr(s1), c_ssize_t(ss), w, h, wt1, o1, log2WD, bd)
            assert np.array_equal(a, b), ("weighted", bd, w, h)
            a = base.copy(); b = base.copy()
            getattr(tab, "put_weighted_bipred" + sfx)(ptr(a), ds, ptr(s1), ptr(s2), ss, w, h, wt1, o1, wt2, o2, log2WD, *extra)
            oracle.o_put_weighted_bipred(ptr(b), c_ssize_t(ds), pb, ptr(s1), ptr(s2), c_ssize_t(ss), w, h, wt1, o1, wt2, o2, log2WD, bd)
            assert np.array_equal(a, b), ("bipred", bd, w, h)


def check_qpel(tab, oracle, quick):
    for bd in _bds(quick):
        rng = XorShift32(0x71E10000 + bd)
        pd, pb = pixel_dtype(bd), (1 if bd <= 8 else 2)
        for (w, h) in ([(8, 4), (16, 16)] if quick else [(8, 4), (4, 8), (16, 12), (24, 32), (64, 64)]):
            S, OS = w + 48, 64
            src = rng.array(S * (h + 16), 0, (1 << bd) - 1, pd)
            sp = ptr_at(src, 6 * S + 16)
            for xf in range(4):
                for yf in range(4):
                    a = np.full(OS * h, 12345, np.int16); b = a.copy()
                    if pb == 1:
                        tab.put_hevc_qpel_8[xf][yf](ptr(a), OS, sp, S, w, h, None)
                    else:
                        tab.put_hevc_qpel_16[xf][yf](ptr(a), OS, sp, S, w, h, None, bd)
                    oracle.o_put_qpel(ptr(b), c_ssize_t(OS), sp, c_ssize_t(S), pb, w, h, xf, yf, bd)
                    assert np.array_equal(a, b), ("qpel", bd, w, h, xf, yf)


def check_epel(tab, oracle, quick):
    for bd in _bds(quick):
        rng = XorShift32(0xE9E10000 + bd)
        pd, pb = pixel_dtype(bd), (1 if bd <= 8 else 2)
        sfx = "_8" if pb == 1 else "_16"
        for (w, h) in ([(2, 4), (8, 8)] if quick else [(2, 4), (4, 2), (8, 8), (6, 8), (12, 16), (32, 32)]):
            S = w + 24
            src = rng.array(S * (h + 8), 0, (1 << bd) - 1, pd)
            sp = ptr_at(src, 3 * S + 4)
            for xf in range(8):
                for yf in range(8):
                    a = np.full((w + 1) * h, 777, np.int16); b = a.copy()
                    # same dispatch as mc_chroma (motion.cc:229-279)
                    if xf == 0 and yf == 0:
                        if pb == 1:
                            tab.put_hevc_epel_8(ptr(a), w + 1, sp, S, w, h, 0, 0, None)
                        else:
                            tab.put_hevc_epel_16(ptr(a), w + 1, sp, S, w, h, 0, 0, None, bd)
                    else:
                        name = "put_hevc_epel_hv" if (xf and yf) else ("put_hevc_epel_h" if xf else "put_hevc_epel_v")
                        getattr(tab, name + sfx)(ptr(a), w + 1, sp, S, w, h, xf, yf, None, bd)
                    oracle.o_put_epel(ptr(b), c_ssize_t(w + 1), sp, c_ssize_t(S), pb, w, h, xf, yf, bd)
                    assert np.array_equal(a, b), ("epel", bd, w, h, xf, yf)


def check_transforms(tab, oracle, quick):
    for bd in _bds(quick):
        pd, pb = pixel_dtype(bd), (1 if bd <= 8 else 2)
        for log2 in [2, 3, 4, 5]:
            rng = XorShift32(0xBEEF1234 + bd * 16 + log2)
            nT, stride = 1 << log2, 48
            for is_dst in ([0, 1] if log2 == 2 else [0]):
                for scenario in range(3):
                    for rep in range(1 if quick else 3):
                        coeffs = coeff_scenarios(rng, nT * nT, scenario)
                        base = rng.array(stride * (nT + 2), 0, (1 << bd) - 1, pd)
                        a = base.copy(); b = base.copy()
                        pa, pbp = ptr_at(a, stride + 3), ptr_at(b, stride + 3)
                        if pb == 1:
                            if is_dst:
                                tab.transform_4x4_dst_add_8(pa, ptr(coeffs), stride)
                                oracle.o_transform_4x4_dst_add_8(pbp, ptr(coeffs), c_ssize_t(stride))
                            else:
                                tab.transform_add_8[log2 - 2](pa, ptr(coeffs), stride)
                                oracle.o_transform_add_8(log2, pbp, ptr(coeffs), c_ssize_t(stride))
                        else:
                            if is_dst:
                                tab.transform_4x4_dst_add_16(pa, ptr(coeffs), stride, bd)
                                oracle.o_transform_4x4_dst_add_16(pbp, ptr(coeffs), c_ssize_t(stride), bd)
                            else:
                                tab.transform_add_16[log2 - 2](pa, ptr(coeffs), stride, bd)
                                oracle.o_transform_add_16(log2, pbp, ptr(coeffs), c_ssize_t(stride), bd)
                        assert np.array_equal(a, b), ("transform_add", bd, log2, is_dst, scenario)
                        # residual-returning variants (transform.cc:283-284: max_coeff_bits 15)
                        ra = np.full(nT * nT, -7, np.int32); rb = ra.copy()
                        if is_dst:
                            tab.transform_idst_4x4(ptr(ra), ptr(coeffs), 20 - bd, 15)
                            oracle.o_transform_idst_4x4(ptr(rb), ptr(coeffs), 20 - bd, 15)
                        else:
                            [tab.transform_idct_4x4, tab.transform_idct_8x8, tab.transform_idct_16x16, tab.transform_idct_32x32][log2 - 2](ptr(ra), ptr(coeffs), 20 - bd, 15)
                            oracle.o_transform_idct(log2, ptr(rb), ptr(coeffs), 20 - bd, 15)
                        assert np.array_equal(ra, rb), ("transform_residual", bd, log2, is_dst, scenario)
                        a = base.copy(); b = base.copy()
                        getattr(tab, "add_residual_8" if pb == 1 else "add_residual_16")(ptr_at(a, stride + 3), stride, ptr(ra), nT, bd)
                        getattr(oracle, "o_add_residual_8" if pb == 1 else "o_add_residual_16")(ptr_at(b, stride + 3), c_ssize_t(stride), ptr(rb), nT, bd)
                        assert np.array_equal(a, b), ("add_residual", bd, log2)
            # skip / rdpcm / bypass / rotate
            coeffs = rng.array(nT * nT, -32768, 32767, np.int16)
            for name, args in [("transform_skip_residual", (5 + log2, 20 - bd)), ("rdpcm_v", (5 + log2, 20 - bd)), ("rdpcm_h", (5 + log2, 20 - bd))]:
                ra = np.zeros(nT * nT, np.int32); rb = ra.copy()
                getattr(tab, name)(ptr(ra), ptr(coeffs), nT, *args)
                getattr(oracle, "o_" + name)(ptr(rb), ptr(coeffs), nT, *args)
                assert np.array_equal(ra, rb), (name, bd, log2)
            for name in ["transform_bypass", "transform_bypass_rdpcm_v", "transform_bypass_rdpcm_h"]:
                ra = np.zeros(nT * nT, np.int32); rb = ra.copy()
                getattr(tab, name)(ptr(ra), ptr(coeffs), nT)
                getattr(oracle, "o_" + name)(ptr(rb), ptr(coeffs), nT)
                assert np.array_equal(ra, rb), (name, bd, log2)
            c1 = coeffs.copy(); c2 = coeffs.copy()
            tab.rotate_coefficients(ptr(c1), nT); oracle.o_rotate_coefficients(ptr(c2), nT)
            assert np.array_equal(c1, c2)
    # 8-bit transform_skip_rdpcm_*_8 = rdpcm residual + add (fallback-dct.cc:94-134)
    rng = XorShift32(0x5C1F0001)
    for log2 in [2, 3, 4, 5]:
        nT, stride = 1 << log2, 40
        coeffs = rng.array(nT * nT, -300, 300, np.int16)
        for name, oname in [("transform_skip_rdpcm_v_8", "o_rdpcm_v"), ("transform_skip_rdpcm_h_8", "o_rdpcm_h")]:
            base = rng.array(stride * (nT + 2), 0, 255, np.uint8)
            a = base.copy(); b = base.copy()
            getattr(tab, name)(ptr_at(a, stride + 3), ptr(coeffs), log2, stride)
            r = np.zeros(nT * nT, np.int32)
            getattr(oracle, oname)(ptr(r), ptr(coeffs), nT, 5 + log2, 12)
            oracle.o_add_residual_8(ptr_at(b, stride + 3), c_ssize_t(stride), ptr(r), nT, 8)
            assert np.array_equal(a, b), (name, log2)


def check_dequant(tab, oracle, quick):
    rng = XorShift32(0xDE0A2222)
    for log2 in [2, 3, 4, 5]:
        n = 1 << (2 * log2)
        for rep in range(2 if quick else 6):
            ncoeff = rng.range(1, n)
            pos = np.array(sorted(set(rng.below(n) for _ in range(ncoeff))), np.int16)
            lvl = rng.array(len(pos), -32768, 32767, np.int16)
            qp = rng.range(0, 51)
            bd = [8, 10, 12][rep % 3]
            fact = [40, 45, 51, 57, 64, 72][qp % 6] << (qp // 6)
            if fact > 32767:
                fact = 32767       # the slot is only used on the int32 path (transform.cc:473-479)
            bdShift = bd + log2 - 5 - 4
            bdShift = max(bdShift, 1)
            a = rng.array(1024, -5, 5, np.int16); b = a.copy()
            tab.dequant_coeff_block(ptr(a), ptr(lvl), ptr(pos), len(pos), fact, 1 << (bdShift - 1), bdShift)
            oracle.o_dequant_coeff_block(ptr(b), ptr(lvl), ptr(pos), len(pos), fact, 1 << (bdShift - 1), bdShift)
            assert np.array_equal(a, b), ("dequant", log2, rep)


def check_deblock(tab, oracle, quick):
    rng = XorShift32(0xDEB10C00)
    stride = 24
    for rep in range(12 if quick else 60):
        base = rng.array(stride * 12, 0, 255, np.uint8)
        if rep % 3 == 0:   # smooth data so the strong/weak paths do something
            base = (base.astype(np.int32) // 16 + 100).astype(np.uint8)
        vertical, dE, dEp, dEq = rng.below(2), 1 + rng.below(2), rng.below(2), rng.below(2)
        tc, fP, fQ = rng.range(0, 24), rng.below(4) != 0, rng.below(4) != 0
        a = base.copy(); b = base.copy()
        tab.deblock_luma_8(ptr_at(a, 5 * stride + 8), stride, vertical, dE, dEp, dEq, tc, int(fP), int(fQ))
        oracle.o_deblock_luma(ptr_at(b, 5 * stride + 8), c_ssize_t(stride), 1, vertical, dE, dEp, dEq, tc, int(fP), int(fQ), 8)
        assert np.array_equal(a, b), ("deblock_luma", rep)
        a = base.copy(); b = base.copy()
        tab.deblock_chroma_8(ptr_at(a, 5 * stride + 8), stride, vertical, tc, int(fP), int(fQ))
        oracle.o_deblock_chroma(ptr_at(b, 5 * stride + 8), c_ssize_t(stride), 1, vertical, tc, int(fP), int(fQ), 8)
        assert np.array_equal(a, b), ("deblock_chroma", rep)


def check_intra(tab, oracle, quick):
    for bd in _bds(quick):
        rng = XorShift32(0x1A7A0000 + bd)
        pd, pb = pixel_dtype(bd), (1 if bd <= 8 else 2)
        sfx = "_8" if pb == 1 else "_16"
        for log2 in [2, 3, 4, 5]:
            nT, stride = 1 << log2, 40
            for cIdx in [0, 1]:
                border = rng.array(4 * nT + 1, 0, (1 << bd) - 1, pd)
                bp = ptr_at(border, 2 * nT)
                base = rng.array(stride * (nT + 2), 0, (1 << bd) - 1, pd)
                a = base.copy(); b = base.copy()
                getattr(tab, "intra_pred_dc" + sfx)(ptr_at(a, stride + 3), stride, nT, cIdx, bp)
                oracle.o_intra_pred_dc(ptr_at(b, stride + 3), c_ssize_t(stride), pb, nT, cIdx, bp)
                assert np.array_equal(a, b), ("dc", bd, nT, cIdx)
                a = base.copy(); b = base.copy()
                getattr(tab, "intra_pred_planar" + sfx)(ptr_at(a, stride + 3), stride, nT, cIdx, bp)
                oracle.o_intra_pred_planar(ptr_at(b, stride + 3), c_ssize_t(stride), pb, nT, cIdx, bp)
                assert np.array_equal(a, b), ("planar", bd, nT, cIdx)
                for mode in (range(2, 35, 3) if quick else range(2, 35)):
                    for dbf in [0, 1]:
                        a = base.copy(); b = base.copy()
                        getattr(tab, "intra_pred_angular" + sfx)(ptr_at(a, stride + 3), stride, bd, dbf, 0, 0, mode, nT, cIdx, bp)
                        oracle.o_intra_pred_angular(ptr_at(b, stride + 3), c_ssize_t(stride), pb, bd, dbf, mode, nT, cIdx, bp)
                        assert np.array_equal(a, b), ("angular", bd, nT, cIdx, mode, dbf)


def check_batch(lib, oracle, quick):
    import ctypes
    rng = XorShift32(0xBA7C4000)
    for bd in [8, 10]:
        pd, pb = pixel_dtype(bd), (1 if bd <= 8 else 2)
        for log2, kind in [(2, 1), (2, 0), (3, 0), (4, 0), (5, 0)]:
            nT, stride, rows = 1 << log2, 256, 64
            nblk = 6
            base = rng.array(stride * rows, 0, (1 << bd) - 1, pd)
            offs = np.array([(i % 3) * 64 + (i // 3) * 32 * stride + 5 for i in range(nblk)], np.int64)
            coeffs = np.concatenate([coeff_scenarios(rng, nT * nT, i % 3) for i in range(nblk)])
            a = base.copy(); b = base.copy()
            rc = lib.lib.m355_transform_add_batch(nblk, log2, kind, bd, ptr(a), ctypes.c_size_t(a.nbytes), ptr(offs), c_ssize_t(stride), ptr(coeffs))
            assert rc == 0, lib.error()
            for i in range(nblk):
                c = coeffs[i * nT * nT:(i + 1) * nT * nT].copy()
                if pb == 1:
                    (oracle.o_transform_4x4_dst_add_8(ptr_at(b, int(offs[i])), ptr(c), c_ssize_t(stride)) if kind else
                     oracle.o_transform_add_8(log2, ptr_at(b, int(offs[i])), ptr(c), c_ssize_t(stride)))
                else:
                    (oracle.o_transform_4x4_dst_add_16(ptr_at(b, int(offs[i])), ptr(c), c_ssize_t(stride), bd) if kind else
                     oracle.o_transform_add_16(log2, ptr_at(b, int(offs[i])), ptr(c), c_ssize_t(stride), bd))
            assert np.array_equal(a, b), ("batch", bd, log2, kind)
    bad = np.array([10 ** 9], np.int64)
    assert lib.lib.m355_transform_add_batch(1, 3, 0, 8, ptr(a), ctypes.c_size_t(a.nbytes), ptr(bad), c_ssize_t(256), ptr(coeffs)) == 3


ALL = [check_weighted, check_qpel, check_epel, check_transforms, check_dequant, check_deblock, check_intra]
