"""Pins the oracle (oracle/hevc_oracle.c) against the REAL reference's scalar fallback functions
(and, where the reference has them, its SSE/AVX2/AVX-512 kernels), slot by slot, using the
reference's own test recipe: identical xorshift32 inputs to both sides, byte-identical whole
strided buffers required (dev-tools/tests.h:29-70, test-transform.cc:61-73, test-intrapred.cc:163-178,
test-deblk.cc, test-dequant.cc:51-58, test-add-residual.cc:52-58)."""
import ctypes
from ctypes import c_int, c_void_p, c_ssize_t

import numpy as np
import pytest

from util import XorShift32, pixel_dtype, ptr, ptr_at

BIT_DEPTHS = [8, 9, 10, 12]


def coeff_scenarios(rng, n, scenario):
    """test-transform.cc:61-73: sparse-small / dense +-2048 / full int16."""
    c = np.zeros(n, np.int16)
    if scenario == 0:
        for _ in range(1 + rng.below(max(1, n // 8))):
            c[rng.below(n)] = rng.range(-512, 512)
    elif scenario == 1:
        c[:] = rng.array(n, -2048, 2048, np.int16)
    else:
        c[:] = rng.array(n, -32768, 32767, np.int16)
    return c


@pytest.mark.parametrize("bd", BIT_DEPTHS)
@pytest.mark.parametrize("log2", [2, 3, 4, 5])
def test_transform_add(oracle, ref, bd, log2):
    rng = XorShift32(0xBEEF1234 + bd * 16 + log2)
    nT, stride = 1 << log2, 48
    pd, pb = pixel_dtype(bd), (1 if bd <= 8 else 2)
    for is_dst in ([0, 1] if log2 == 2 else [0]):
        for scenario in range(3):
            for rep in range(6):
                coeffs = coeff_scenarios(rng, nT * nT, scenario)
                base = rng.array(stride * (nT + 2), 0, (1 << bd) - 1, pd)
                outs = []
                for simd in (0, 1):
                    if simd and bd > 8:
                        continue
                    d = base.copy()
                    ref.ref_transform_add(simd, log2, is_dst, ptr_at(d, stride + 3), pb, ptr(coeffs), c_ssize_t(stride), bd)
                    outs.append(d)
                d = base.copy()
                if pb == 1:
                    if is_dst:
                        oracle.o_transform_4x4_dst_add_8(ptr_at(d, stride + 3), ptr(coeffs), c_ssize_t(stride))
                    else:
                        oracle.o_transform_add_8(log2, ptr_at(d, stride + 3), ptr(coeffs), c_ssize_t(stride))
                else:
                    if is_dst:
                        oracle.o_transform_4x4_dst_add_16(ptr_at(d, stride + 3), ptr(coeffs), c_ssize_t(stride), bd)
                    else:
                        oracle.o_transform_add_16(log2, ptr_at(d, stride + 3), ptr(coeffs), c_ssize_t(stride), bd)
                for o in outs:
                    assert np.array_equal(o, d), (bd, log2, is_dst, scenario)


@pytest.mark.parametrize("log2", [2, 3, 4, 5])
def test_transform_residual(oracle, ref, log2):
    rng = XorShift32(0x1D0C7 + log2)
    nT = 1 << log2
    for is_dst in ([0, 1] if log2 == 2 else [0]):
        for bdShift in (12, 10, 8):
            for mcb in (15, 16) if bdShift == 12 else (15,):
                for scenario in range(3):
                    coeffs = coeff_scenarios(rng, nT * nT, scenario)
                    a = np.zeros(nT * nT, np.int32)
                    b = np.zeros(nT * nT, np.int32)
                    ref.ref_transform_residual(0, log2, is_dst, ptr(a), ptr(coeffs), bdShift, mcb)
                    if is_dst:
                        oracle.o_transform_idst_4x4(ptr(b), ptr(coeffs), bdShift, mcb)
                    else:
                        oracle.o_transform_idct(log2, ptr(b), ptr(coeffs), bdShift, mcb)
                    assert np.array_equal(a, b)


@pytest.mark.parametrize("bd", [8, 9, 10, 12, 16])
def test_add_residual(oracle, ref, bd):
    """test-add-residual.cc: nT 4..32, 4 clip scenarios."""
    rng = XorShift32(0x00C0FFEE + bd)
    pd, pb = pixel_dtype(bd), (1 if bd <= 8 else 2)
    maxv = (1 << bd) - 1
    for nT in (4, 8, 16, 32):
        for scen in range(4):
            stride = nT + 16
            if scen == 0:
                base = rng.array(stride * nT, 0, maxv, pd); r = rng.array(nT * nT, -maxv, maxv, np.int32)
            elif scen == 1:
                base = np.full(stride * nT, maxv, pd); r = rng.array(nT * nT, 0, 70000, np.int32)
            elif scen == 2:
                base = np.zeros(stride * nT, pd); r = rng.array(nT * nT, -70000, 0, np.int32)
            else:
                base = rng.array(stride * nT, 0, maxv, pd); r = rng.array(nT * nT, -3, 3, np.int32)
            a, b, c = base.copy(), base.copy(), base.copy()
            ref.ref_add_residual(0, ptr(a), pb, c_ssize_t(stride), ptr(r), nT, bd)
            ref.ref_add_residual(1, ptr(c), pb, c_ssize_t(stride), ptr(r), nT, bd)
            (oracle.o_add_residual_8 if pb == 1 else oracle.o_add_residual_16)(ptr(b), c_ssize_t(stride), ptr(r), nT, bd)
            assert np.array_equal(a, b) and np.array_equal(c, b)


def test_dequant(oracle, ref):
    """test-dequant.cc: qP 0..51 step 3, 8 sparsities; fact<=32767 int32 contract."""
    rng = XorShift32(0xD2C0FFEE)
    level_scale = [40, 45, 51, 57, 64, 72]
    for log2 in (2, 3, 4, 5):
        n = 1 << (2 * log2)
        for qp in range(0, 52, 3):
            for nnz in sorted({1, 2, n // 64 or 1, n // 16, n // 8, n // 4, n // 2, n}):
                fact = level_scale[qp % 6] << (qp // 6)
                bdShift = 8 + log2 - 5 - 4
                if bdShift < 1:
                    bdShift = 1
                offset = 1 << (bdShift - 1)
                pos = np.array(sorted(set(rng.below(n) for _ in range(nnz))), np.int16)
                lvl = rng.array(len(pos), -32768, 32767, np.int16)
                if fact > 32767:
                    continue
                outs = []
                for f in (lambda buf: ref.ref_dequant_coeff_block(0, ptr(buf), ptr(lvl), ptr(pos), len(pos), fact, offset, bdShift),
                          lambda buf: ref.ref_dequant_coeff_block(1, ptr(buf), ptr(lvl), ptr(pos), len(pos), fact, offset, bdShift),
                          lambda buf: oracle.o_dequant_coeff_block(ptr(buf), ptr(lvl), ptr(pos), len(pos), fact, offset, bdShift)):
                    buf = np.zeros(n, np.int16)
                    f(buf)
                    outs.append(buf)
                assert np.array_equal(outs[0], outs[2]) and np.array_equal(outs[1], outs[2])


def test_residual_misc(oracle, ref):
    rng = XorShift32(0x5A1F0001)
    names = ["o_transform_skip_residual", "o_rdpcm_v", "o_rdpcm_h", "o_transform_bypass",
             "o_transform_bypass_rdpcm_v", "o_transform_bypass_rdpcm_h"]
    for log2 in (2, 3, 4, 5):
        nT = 1 << log2
        for bd in (8, 10, 12):
            for which in range(6):
                coeffs = rng.array(nT * nT, -32768, 32767, np.int16)
                a = np.zeros(nT * nT, np.int32); b = np.zeros(nT * nT, np.int32)
                tsShift, bdShift = 5 + log2, 20 - bd
                ref.ref_residual_misc(which, ptr(a), ptr(coeffs), nT, tsShift, bdShift)
                if which < 3:
                    getattr(oracle, names[which])(ptr(b), ptr(coeffs), nT, tsShift, bdShift)
                else:
                    getattr(oracle, names[which])(ptr(b), ptr(coeffs), nT)
                assert np.array_equal(a, b), (log2, bd, which)
        c1 = rng.array(nT * nT, -32768, 32767, np.int16); c2 = c1.copy()
        ref.ref_rotate_coefficients(ptr(c1), nT); oracle.o_rotate_coefficients(ptr(c2), nT)
        assert np.array_equal(c1, c2)


PB_SIZES = [(8, 4), (4, 8), (8, 8), (16, 4), (4, 16), (16, 12), (12, 16), (16, 16), (24, 32), (32, 8), (64, 64), (48, 64)]


@pytest.mark.parametrize("bd", BIT_DEPTHS)
def test_qpel(oracle, ref, bd):
    """Not pinned by any reference test: all 16 phases, 8/9/10/12-bit; the reference's
    SSE kernels (8-bit, aligned 64-sample output rows as in motion.cc:331) are checked as well."""
    rng = XorShift32(0x71E10000 + bd)
    pd, pb = pixel_dtype(bd), (1 if bd <= 8 else 2)
    OS = 64
    for (w, h) in PB_SIZES:
        S = w + 48
        src = rng.array(S * (h + 16), 0, (1 << bd) - 1, pd)
        sp = ptr_at(src, 6 * S + 16)
        for xf in range(4):
            for yf in range(4):
                a = np.full(OS * h, 12345, np.int16); b = a.copy()
                ref.ref_put_qpel(0, ptr(a), c_ssize_t(OS), sp, c_ssize_t(S), pb, w, h, xf, yf, bd)
                oracle.o_put_qpel(ptr(b), c_ssize_t(OS), sp, c_ssize_t(S), pb, w, h, xf, yf, bd)
                assert np.array_equal(a, b), (w, h, xf, yf)
                if bd == 8 and w % 8 == 0:
                    c = np.full(OS * h, 12345, np.int16)
                    ref.ref_put_qpel(1, ptr(c), c_ssize_t(OS), sp, c_ssize_t(S), pb, w, h, xf, yf, bd)
                    assert np.array_equal(c.reshape(h, OS)[:, :w], b.reshape(h, OS)[:, :w]), (w, h, xf, yf)


@pytest.mark.parametrize("bd", BIT_DEPTHS)
def test_epel(oracle, ref, bd):
    rng = XorShift32(0xE9E10000 + bd)
    pd, pb = pixel_dtype(bd), (1 if bd <= 8 else 2)
    for (w, h) in [(2, 4), (4, 2), (4, 4), (8, 2), (8, 8), (6, 8), (12, 16), (16, 16), (32, 32), (24, 32)]:
        S = w + 24
        src = rng.array(S * (h + 8), 0, (1 << bd) - 1, pd)
        for xf in range(8):
            for yf in range(8):
                a = np.full((w + 1) * h, 777, np.int16); b = a.copy()
                ref.ref_put_epel(0, ptr(a), c_ssize_t(w + 1), ptr_at(src, 3 * S + 4), c_ssize_t(S), pb, w, h, xf, yf, bd)
                oracle.o_put_epel(ptr(b), c_ssize_t(w + 1), ptr_at(src, 3 * S + 4), c_ssize_t(S), pb, w, h, xf, yf, bd)
                assert np.array_equal(a, b), (w, h, xf, yf)


@pytest.mark.parametrize("bd", BIT_DEPTHS)
def test_weighted_prediction(oracle, ref, bd):
    rng = XorShift32(0x3E167ED0 + bd)
    pd, pb = pixel_dtype(bd), (1 if bd <= 8 else 2)
    for (w, h) in [(4, 4), (8, 4), (16, 16), (12, 16), (64, 64), (2, 2), (6, 8)]:
        ss, ds = 64, w + 5
        s1 = rng.array(ss * h, -(1 << 14), (1 << 14) - 1, np.int16)
        s2 = rng.array(ss * h, -(1 << 14), (1 << 14) - 1, np.int16)
        base = rng.array(ds * h, 0, (1 << bd) - 1, pd)
        for rep in range(4):
            denom = rng.below(8)
            log2WD = denom + max(2, 14 - bd)
            w1 = (1 << denom) + rng.range(-128, 127); w2 = (1 << denom) + rng.range(-128, 127)
            o1 = rng.range(-128, 127) << (bd - 8); o2 = rng.range(-128, 127) << (bd - 8)
            for simd in ((0, 1) if bd == 8 else (0,)):
                a, b = base.copy(), base.copy()
                ref.ref_put_unweighted_pred(simd, ptr(a), c_ssize_t(ds), pb, ptr(s1), c_ssize_t(ss), w, h, bd)
                oracle.o_put_unweighted_pred(ptr(b), c_ssize_t(ds), pb, ptr(s1), c_ssize_t(ss), w, h, bd)
                assert np.array_equal(a, b)
                a, b = base.copy(), base.copy()
                ref.ref_put_weighted_pred_avg(simd, ptr(a), c_ssize_t(ds), pb, ptr(s1), ptr(s2), c_ssize_t(ss), w, h, bd)
                oracle.o_put_weighted_pred_avg(ptr(b), c_ssize_t(ds), pb, ptr(s1), ptr(s2), c_ssize_t(ss), w, h, bd)
                assert np.array_equal(a, b)
            a, b = base.copy(), base.copy()
            ref.ref_put_weighted_pred(0, ptr(a), c_ssize_t(ds), pb, ptr(s1), c_ssize_t(ss), w, h, w1, o1, log2WD, bd)
            oracle.o_put_weighted_pred(ptr(b), c_ssize_t(ds), pb, ptr(s1), c_ssize_t(ss), w, h, w1, o1, log2WD, bd)
            assert np.array_equal(a, b)
            a, b = base.copy(), base.copy()
            ref.ref_put_weighted_bipred(0, ptr(a), c_ssize_t(ds), pb, ptr(s1), ptr(s2), c_ssize_t(ss), w, h, w1, o1, w2, o2, log2WD, bd)
            oracle.o_put_weighted_bipred(ptr(b), c_ssize_t(ds), pb, ptr(s1), ptr(s2), c_ssize_t(ss), w, h, w1, o1, w2, o2, log2WD, bd)
            assert np.array_equal(a, b)


@pytest.mark.parametrize("bd", [8, 10, 12])
def test_intra_predictors(oracle, ref, bd):
    """test-intrapred.cc:163-178: all 35 modes x nT x cIdx x disableBoundaryFilter."""
    rng = XorShift32(0x1234ABCD + bd)
    pd, pb = pixel_dtype(bd), (1 if bd <= 8 else 2)
    for nT in (4, 8, 16, 32):
        stride = nT + 7
        for cIdx in (0, 1):
            for disable in (0, 1):
                for mode in range(35):
                    border = rng.array(4 * 64 + 1, 0, (1 << bd) - 1, pd)
                    which = 0 if mode == 0 else (1 if mode == 1 else 2)
                    outs = []
                    for simd in ((0, 1) if bd == 8 else (0,)):
                        d = np.full(stride * nT, 3, pd)
                        ref.ref_intra_pred(simd, which, ptr(d), c_ssize_t(stride), pb, bd, disable, mode, nT, cIdx, ptr_at(border, 128))
                        outs.append(d)
                    d = np.full(stride * nT, 3, pd)
                    if which == 0:
                        oracle.o_intra_pred_planar(ptr(d), c_ssize_t(stride), pb, nT, cIdx, ptr_at(border, 128))
                    elif which == 1:
                        oracle.o_intra_pred_dc(ptr(d), c_ssize_t(stride), pb, nT, cIdx, ptr_at(border, 128))
                    else:
                        oracle.o_intra_pred_angular(ptr(d), c_ssize_t(stride), pb, bd, disable, mode, nT, cIdx, ptr_at(border, 128))
                    for o in outs:
                        assert np.array_equal(o, d), (nT, cIdx, disable, mode)


@pytest.mark.parametrize("bd", [8, 10])
def test_intra_sample_filtering(oracle, ref, bd):
    rng = XorShift32(0xF117E2 + bd)
    pd, pb = pixel_dtype(bd), (1 if bd <= 8 else 2)
    for nT in (4, 8, 16, 32):
        for cIdx in (0, 1):
            for strong in (0, 1):
                for mode in range(35):
                    for smooth in (0, 1):
                        if smooth:  # nearly flat border so the strong (bilinear) path triggers
                            base = rng.range(16, (1 << bd) - 17)
                            border = np.array([base + rng.range(-1, 1) for _ in range(257)], pd)
                        else:
                            border = rng.array(257, 0, (1 << bd) - 1, pd)
                        a, b = border.copy(), border.copy()
                        ref.ref_intra_sample_filtering(ptr_at(a, 128), pb, nT, cIdx, mode, strong, bd)
                        oracle.o_intra_sample_filtering(ptr_at(b, 128), pb, nT, cIdx, mode, strong, bd)
                        assert np.array_equal(a, b), (nT, cIdx, strong, mode, smooth)


@pytest.mark.parametrize("bd", [8, 10])
def test_deblock_kernels(oracle, ref, bd):
    """test-deblk.cc: all dE/dEp/dEq/filterP/filterQ, tc in [1,25]; chroma too (not pinned upstream)."""
    rng = XorShift32(0xDEB10C + bd)
    pd, pb = pixel_dtype(bd), (1 if bd <= 8 else 2)
    stride = 24
    for vertical in (0, 1):
        for dE in (1, 2):
            for dEp in (0, 1):
                for dEq in (0, 1):
                    for fP in (0, 1):
                        for fQ in (0, 1):
                            for rep in range(12):
                                tc = rng.range(1, 25) << (bd - 8)
                                if rep % 3 == 0:
                                    base = rng.array(stride * 16, 0, (1 << bd) - 1, pd)
                                else:  # smooth-ish so the weak filter's |delta|<10tc branch is hit
                                    m = rng.range(30, (1 << bd) - 31)
                                    base = np.array([m + rng.range(-12, 12) for _ in range(stride * 16)], pd)
                                outs = []
                                for simd in ((0, 1) if bd == 8 else (0,)):
                                    d = base.copy()
                                    ref.ref_deblock_luma(simd, ptr_at(d, 6 * stride + 8), c_ssize_t(stride), pb, vertical, dE, dEp, dEq, tc, fP, fQ, bd)
                                    outs.append(d)
                                d = base.copy()
                                oracle.o_deblock_luma(ptr_at(d, 6 * stride + 8), c_ssize_t(stride), pb, vertical, dE, dEp, dEq, tc, fP, fQ, bd)
                                for o in outs:
                                    assert np.array_equal(o, d)
                                a, b = base.copy(), base.copy()
                                ref.ref_deblock_chroma(0, ptr_at(a, 6 * stride + 8), c_ssize_t(stride), pb, vertical, tc, fP, fQ, bd)
                                oracle.o_deblock_chroma(ptr_at(b, 6 * stride + 8), c_ssize_t(stride), pb, vertical, tc, fP, fQ, bd)
                                assert np.array_equal(a, b)
