/*
 * hevc_oracle.c — CPU restatement of libde265's pixel-reconstruction path (plain C99).
 *
 * TEST INFRASTRUCTURE ONLY: see hevc_oracle.h.  Parity pinned against the reference's own scalar
 * fallback functions (slot level) and the girlshy golden MD5 (picture level).
 * Citations are into /root/reference/libde265/.
 */
#include "hevc_oracle.h"

#include <stdlib.h>
#include <string.h>

/* ---------------------------------------------------------------- helpers (util.h:78-126) ----- */

static inline int clip3(int lo, int hi, int v) { return v < lo ? lo : (v > hi ? hi : v); }
static inline int clip_bd(int v, int bd) { return clip3(0, (1 << bd) - 1, v); }
static inline int iabs(int v) { return v < 0 ? -v : v; }
static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }
static inline int isign(int v) { return (v > 0) - (v < 0); }

static inline int px_get(const void* p, int pixel_bytes, ptrdiff_t idx) {
  return pixel_bytes == 1 ? ((const uint8_t*)p)[idx] : ((const uint16_t*)p)[idx];
}
static inline void px_set(void* p, int pixel_bytes, ptrdiff_t idx, int v) {
  if (pixel_bytes == 1) ((uint8_t*)p)[idx] = (uint8_t)v; else ((uint16_t*)p)[idx] = (uint16_t)v;
}

/* ------------------------------------------------------------------ transform matrices -------- */

/* The 32-point HEVC core transform (fallback-dct.cc:512-545 mat_dct) is M[k][n] = c(k*(2n+1)) with
 * c(m) the quarter-wave table below, period 128, c(m) = -c(64-m) = -c(64+m) = c(128-m).  The tests
 * compare every derived entry against the reference through the fallback transforms. */
static const int8_t dct_quarter_wave[33] = {64, 90, 90, 90, 89, 88, 87, 85, 83, 82, 80, 78, 75, 73, 70, 67, 64,
                                            61, 57, 54, 50, 46, 43, 38, 36, 31, 25, 22, 18, 13, 9,  4,  0};
static int8_t g_dct[32][32];
static int g_dct_ready = 0;

static int dct_wave(int m) {
  m &= 127;
  if (m <= 32) return dct_quarter_wave[m];
  if (m <= 64) return -dct_quarter_wave[64 - m];
  if (m < 96) return -dct_quarter_wave[m - 64];
  return dct_quarter_wave[128 - m];
}
static void dct_init(void) {
  if (g_dct_ready) return;
  for (int k = 0; k < 32; k++)
    for (int n = 0; n < 32; n++) g_dct[k][n] = (int8_t)dct_wave(k * (2 * n + 1));
  g_dct_ready = 1;
}

/* fallback-dct.cc:260-265 mat_8_357 (4x4 DST-VII) */
static const int8_t g_dst[4][4] = {{29, 55, 74, 84}, {74, 74, 0, -74}, {84, -29, -74, 55}, {55, -84, 74, -29}};

/* Two-stage inverse transform to a residual block (A2 in SURVEY appendix):
 *   g = clip(CoeffMin,CoeffMax, (sum_j M[j][i]*c[j][col] + 64) >> 7)       columns first
 *   r = (sum_j M[j][i]*g[row][j] + (1<<(bdShift-1))) >> bdShift           rows, NOT clipped
 * fallback-dct.cc:550-691 (add variant, clip to int16) and :695-836 (residual variant). The
 * reference's last-nonzero pruning (:614-617) only skips zero terms, so it is omitted here. */
static void inv_transform(int32_t* r, const int16_t* coeffs, int nT, int is_dst, int bdShift, int cmin, int cmax) {
  int16_t g[32 * 32];
  dct_init();
  const int fact = 32 / nT;
  const int rnd1 = 1 << 6, rnd2 = 1 << (bdShift - 1);
  for (int c = 0; c < nT; c++)
    for (int i = 0; i < nT; i++) {
      int sum = 0;
      for (int j = 0; j < nT; j++) {
        int m = is_dst ? g_dst[j][i] : g_dct[fact * j][i];
        sum += m * coeffs[c + j * nT];
      }
      g[c + i * nT] = (int16_t)clip3(cmin, cmax, (sum + rnd1) >> 7);
    }
  for (int y = 0; y < nT; y++)
    for (int i = 0; i < nT; i++) {
      int sum = 0;
      for (int j = 0; j < nT; j++) {
        int m = is_dst ? g_dst[j][i] : g_dct[fact * j][i];
        sum += m * g[y * nT + j];
      }
      r[y * nT + i] = (sum + rnd2) >> bdShift;
    }
}

static void add_res(void* dst, ptrdiff_t stride, int pixel_bytes, const int32_t* r, int nT, int bd) {
  for (int y = 0; y < nT; y++)
    for (int x = 0; x < nT; x++)
      px_set(dst, pixel_bytes, y * stride + x, clip_bd(px_get(dst, pixel_bytes, y * stride + x) + r[y * nT + x], bd));
}

void o_transform_add_8(int log2nT, uint8_t* dst, const int16_t* coeffs, ptrdiff_t stride) {
  int32_t r[32 * 32];
  inv_transform(r, coeffs, 1 << log2nT, 0, 20 - 8, -32768, 32767);
  add_res(dst, stride, 1, r, 1 << log2nT, 8);
}
void o_transform_add_16(int log2nT, uint16_t* dst, const int16_t* coeffs, ptrdiff_t stride, int bit_depth) {
  int32_t r[32 * 32];
  inv_transform(r, coeffs, 1 << log2nT, 0, 20 - bit_depth, -32768, 32767);
  add_res(dst, stride, 2, r, 1 << log2nT, bit_depth);
}
/* fallback-dct.cc:269-336: the DST variant additionally clips `out` to int16 before the add, which
 * cannot change Clip_BitDepth(dst+out) for bit depths <= 15. Restated literally. */
void o_transform_4x4_dst_add_8(uint8_t* dst, const int16_t* coeffs, ptrdiff_t stride) {
  int32_t r[16];
  inv_transform(r, coeffs, 4, 1, 20 - 8, -32768, 32767);
  for (int i = 0; i < 16; i++) r[i] = clip3(-32768, 32767, r[i]);
  add_res(dst, stride, 1, r, 4, 8);
}
void o_transform_4x4_dst_add_16(uint16_t* dst, const int16_t* coeffs, ptrdiff_t stride, int bit_depth) {
  int32_t r[16];
  inv_transform(r, coeffs, 4, 1, 20 - bit_depth, -32768, 32767);
  for (int i = 0; i < 16; i++) r[i] = clip3(-32768, 32767, r[i]);
  add_res(dst, stride, 2, r, 4, bit_depth);
}
void o_transform_idst_4x4(int32_t* dst, const int16_t* coeffs, int bdShift, int max_coeff_bits) {
  inv_transform(dst, coeffs, 4, 1, bdShift, -(1 << max_coeff_bits), (1 << max_coeff_bits) - 1);
}
void o_transform_idct(int log2nT, int32_t* dst, const int16_t* coeffs, int bdShift, int max_coeff_bits) {
  inv_transform(dst, coeffs, 1 << log2nT, 0, bdShift, -(1 << max_coeff_bits), (1 << max_coeff_bits) - 1);
}
void o_add_residual_8(uint8_t* dst, ptrdiff_t stride, const int32_t* r, int nT, int bit_depth) {
  add_res(dst, stride, 1, r, nT, bit_depth);
}
void o_add_residual_16(uint16_t* dst, ptrdiff_t stride, const int32_t* r, int nT, int bit_depth) {
  add_res(dst, stride, 2, r, nT, bit_depth);
}

void o_dequant_coeff_block(int16_t* coeffBuf, const int16_t* coeffList, const int16_t* coeffPos, int nCoeff,
                           int32_t fact, int32_t offset, int32_t bdShift) {
  for (int i = 0; i < nCoeff; i++)
    coeffBuf[coeffPos[i]] = (int16_t)clip3(-32768, 32767, (coeffList[i] * fact + offset) >> bdShift);
}

void o_transform_skip_residual(int32_t* residual, const int16_t* coeffs, int nT, int tsShift, int bdShift) {
  const int rnd = 1 << (bdShift - 1);
  for (int i = 0; i < nT * nT; i++) {
    int32_t c = (int32_t)((uint32_t)(int32_t)coeffs[i] << tsShift);
    residual[i] = (c + rnd) >> bdShift;
  }
}
void o_rdpcm_v(int32_t* residual, const int16_t* coeffs, int nT, int tsShift, int bdShift) {
  const int rnd = 1 << (bdShift - 1);
  for (int x = 0; x < nT; x++) {
    int sum = 0;
    for (int y = 0; y < nT; y++) {
      int32_t c = (int32_t)((uint32_t)(int32_t)coeffs[x + y * nT] << tsShift);
      sum += (c + rnd) >> bdShift;
      residual[y * nT + x] = sum;
    }
  }
}
void o_rdpcm_h(int32_t* residual, const int16_t* coeffs, int nT, int tsShift, int bdShift) {
  const int rnd = 1 << (bdShift - 1);
  for (int y = 0; y < nT; y++) {
    int sum = 0;
    for (int x = 0; x < nT; x++) {
      int32_t c = (int32_t)((uint32_t)(int32_t)coeffs[x + y * nT] << tsShift);
      sum += (c + rnd) >> bdShift;
      residual[y * nT + x] = sum;
    }
  }
}
void o_transform_bypass(int32_t* r, const int16_t* coeffs, int nT) {
  for (int i = 0; i < nT * nT; i++) r[i] = coeffs[i];
}
void o_transform_bypass_rdpcm_v(int32_t* r, const int16_t* coeffs, int nT) {
  for (int x = 0; x < nT; x++) {
    int sum = 0;
    for (int y = 0; y < nT; y++) { sum += coeffs[x + y * nT]; r[y * nT + x] = sum; }
  }
}
void o_transform_bypass_rdpcm_h(int32_t* r, const int16_t* coeffs, int nT) {
  for (int y = 0; y < nT; y++) {
    int sum = 0;
    for (int x = 0; x < nT; x++) { sum += coeffs[x + y * nT]; r[y * nT + x] = sum; }
  }
}
void o_rotate_coefficients(int16_t* coeff, int nT) {
  for (int i = 0; i < nT * nT / 2; i++) {
    int16_t t = coeff[i];
    coeff[i] = coeff[nT * nT - 1 - i];
    coeff[nT * nT - 1 - i] = t;
  }
}

/* --------------------------------------------------------------- interpolation (A3) ---------- */

/* fallback-motion.cc:531,543,555 — taps over x-3..x+4; the 1/4 and 3/4 filters have 7 taps */
static const int8_t qpel_taps[4][8] = {{0, 0, 0, 64, 0, 0, 0, 0},
                                       {-1, 4, -10, 58, 17, -5, 1, 0},
                                       {-1, 4, -11, 40, 40, -11, 4, -1},
                                       {0, 1, -5, 17, 58, -10, 4, -1}};
/* fallback-motion.cc:357-364 — taps over x-1..x+2 */
static const int8_t epel_taps[8][4] = {{0, 64, 0, 0},   {-2, 58, 10, -2}, {-4, 54, 16, -2}, {-6, 46, 28, -4},
                                       {-4, 36, 36, -4}, {-4, 28, 46, -6}, {-2, 16, 54, -4}, {-2, 10, 58, -2}};

/* Generic separable interpolation with the reference's exact truncation points:
 *  H:  t = (frac ? sum(taps*ref) >> shift1 : ref)             stored as int16 (mcbuffer)
 *  V:  out = (frac ? sum(taps*t) >> (xFrac ? 6 : shift1) : t)  stored as int16
 * full-pel both ways: out = ref << max(2,14-bd)  (fallback-motion.cc:431-485, 262-302). */
static void interp(int16_t* out, ptrdiff_t out_stride, const void* src, ptrdiff_t ss, int pb, int W, int H,
                   int xf, int yf, int bd, int ntaps, int before) {
  const int shift1 = bd - 8;
  if (xf == 0 && yf == 0) {
    const int shift3 = imax(2, 14 - bd);
    for (int y = 0; y < H; y++)
      for (int x = 0; x < W; x++) out[y * out_stride + x] = (int16_t)(px_get(src, pb, y * ss + x) << shift3);
    return;
  }
  int16_t* tmp = (int16_t*)malloc(sizeof(int16_t) * (size_t)W * (size_t)(H + ntaps));
  /* rows actually needed: the reference reads no extra rows when yFrac==0 (extra_top/bottom = 0) */
  int k0 = 0, k1 = ntaps - 1;
  if (ntaps == 8 && yf == 1) k1 = 6;
  if (ntaps == 8 && yf == 3) k0 = 1;
  const int yy0 = yf ? k0 : before, yy1 = yf ? H + k1 : before + H;
  for (int yy = yy0; yy < yy1; yy++) {
    const int y = yy - before;
    for (int x = 0; x < W; x++) {
      int v;
      if (xf == 0) v = px_get(src, pb, y * ss + x);
      else {
        int s = 0;
        for (int k = 0; k < ntaps; k++) {
          int tap = ntaps == 8 ? qpel_taps[xf][k] : epel_taps[xf][k];
          if (tap == 0) continue; /* 7-tap phases: the reference never touches that sample */
          s += tap * px_get(src, pb, y * ss + x + k - before);
        }
        v = s >> shift1;
      }
      tmp[yy * W + x] = (int16_t)v;
    }
  }
  const int vshift = (xf == 0) ? shift1 : 6;
  for (int y = 0; y < H; y++)
    for (int x = 0; x < W; x++) {
      int v;
      if (yf == 0) v = tmp[(y + before) * W + x];
      else {
        int s = 0;
        for (int k = 0; k < ntaps; k++) {
          int tap = ntaps == 8 ? qpel_taps[yf][k] : epel_taps[yf][k];
          if (tap == 0) continue;
          s += tap * tmp[(y + k) * W + x];
        }
        v = s >> vshift;
      }
      out[y * out_stride + x] = (int16_t)v;
    }
  free(tmp);
}

void o_put_qpel(int16_t* out, ptrdiff_t out_stride, const void* src, ptrdiff_t srcstride, int pixel_bytes, int nPbW,
                int nPbH, int xFracL, int yFracL, int bit_depth) {
  interp(out, out_stride, src, srcstride, pixel_bytes, nPbW, nPbH, xFracL, yFracL, bit_depth, 8, 3);
}
void o_put_epel(int16_t* out, ptrdiff_t out_stride, const void* src, ptrdiff_t srcstride, int pixel_bytes, int nPbWC,
                int nPbHC, int xFracC, int yFracC, int bit_depth) {
  interp(out, out_stride, src, srcstride, pixel_bytes, nPbWC, nPbHC, xFracC, yFracC, bit_depth, 4, 1);
}

/* ------------------------------------------------------- prediction write-back (A4) ---------- */

void o_put_unweighted_pred(void* dst, ptrdiff_t ds, int pb, const int16_t* src, ptrdiff_t ss, int w, int h, int bd) {
  const int shift1 = imax(2, 14 - bd), off = 1 << (shift1 - 1);
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) px_set(dst, pb, y * ds + x, clip_bd((src[y * ss + x] + off) >> shift1, bd));
}
void o_put_weighted_pred_avg(void* dst, ptrdiff_t ds, int pb, const int16_t* s1, const int16_t* s2, ptrdiff_t ss, int w,
                             int h, int bd) {
  const int shift2 = imax(3, 15 - bd), off = 1 << (shift2 - 1);
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++)
      px_set(dst, pb, y * ds + x, clip_bd((s1[y * ss + x] + s2[y * ss + x] + off) >> shift2, bd));
}
void o_put_weighted_pred(void* dst, ptrdiff_t ds, int pb, const int16_t* src, ptrdiff_t ss, int w, int h, int wt, int o,
                         int log2WD, int bd) {
  const int rnd = 1 << (log2WD - 1);
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) px_set(dst, pb, y * ds + x, clip_bd(((src[y * ss + x] * wt + rnd) >> log2WD) + o, bd));
}
void o_put_weighted_bipred(void* dst, ptrdiff_t ds, int pb, const int16_t* s1, const int16_t* s2, ptrdiff_t ss, int w,
                           int h, int w1, int o1, int w2, int o2, int log2WD, int bd) {
  const int rnd = (int)((unsigned)(o1 + o2 + 1) << log2WD);
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++)
      px_set(dst, pb, y * ds + x, clip_bd((s1[y * ss + x] * w1 + s2[y * ss + x] * w2 + rnd) >> (log2WD + 1), bd));
}

/* ---------------------------------------------------------------- intra prediction (A5) ------- */

/* intrapred.cc:268-274 */
static const int8_t intra_angle[35] = {0,   0,   32,  26,  21,  17, 13, 9,  5, 2, 0, -2, -5, -9, -13, -17, -21, -26,
                                       -32, -26, -21, -17, -13, -9, -5, -2, 0, 2, 5, 9,  13, 17, 21,  26,  32};
static const int16_t intra_inv_angle[15] = {-4096, -1638, -910, -630, -482, -390, -315, -256,
                                            -315,  -390,  -482, -630, -910, -1638, -4096};

#define B(i) px_get(border, pb, (i))

void o_intra_pred_planar(void* dst, ptrdiff_t stride, int pb, int nT, int cIdx, const void* border) {
  (void)cIdx;
  int log2 = 0;
  while ((1 << log2) < nT) log2++;
  for (int y = 0; y < nT; y++)
    for (int x = 0; x < nT; x++)
      px_set(dst, pb, x + y * stride,
             ((nT - 1 - x) * B(-1 - y) + (x + 1) * B(1 + nT) + (nT - 1 - y) * B(1 + x) + (y + 1) * B(-1 - nT) + nT) >>
                 (log2 + 1));
}
void o_intra_pred_dc(void* dst, ptrdiff_t stride, int pb, int nT, int cIdx, const void* border) {
  int log2 = 0;
  while ((1 << log2) < nT) log2++;
  int dc = 0;
  for (int i = 0; i < nT; i++) dc += B(i + 1) + B(-i - 1);
  dc = (dc + nT) >> (log2 + 1);
  for (int y = 0; y < nT; y++)
    for (int x = 0; x < nT; x++) px_set(dst, pb, x + y * stride, dc);
  if (cIdx == 0 && nT < 32) {
    px_set(dst, pb, 0, (B(-1) + 2 * dc + B(1) + 2) >> 2);
    for (int x = 1; x < nT; x++) px_set(dst, pb, x, (B(x + 1) + 3 * dc + 2) >> 2);
    for (int y = 1; y < nT; y++) px_set(dst, pb, y * stride, (B(-y - 1) + 3 * dc + 2) >> 2);
  }
}
void o_intra_pred_angular(void* dst, ptrdiff_t stride, int pb, int bit_depth, int disableBoundaryFilter, int mode,
                          int nT, int cIdx, const void* border) {
  int refm[4 * 64 + 1];
  int* ref = refm + 2 * 64;
  const int angle = intra_angle[mode];
  const int sgn = (mode >= 18) ? 1 : -1; /* mode>=18 walks the top row (+i), else the left column (-i) */
  for (int x = 0; x <= nT; x++) ref[x] = B(sgn * x);
  if (angle < 0) {
    const int inv = intra_inv_angle[mode - 11];
    if (((nT * angle) >> 5) < -1)
      for (int x = (nT * angle) >> 5; x <= -1; x++) ref[x] = B(-sgn * ((x * inv + 128) >> 8));
  } else {
    for (int x = nT + 1; x <= 2 * nT; x++) ref[x] = B(sgn * x);
  }
  for (int y = 0; y < nT; y++)
    for (int x = 0; x < nT; x++) {
      const int a = (mode >= 18) ? y : x; /* coordinate along the prediction direction */
      const int b = (mode >= 18) ? x : y;
      const int iIdx = ((a + 1) * angle) >> 5, iFact = ((a + 1) * angle) & 31;
      int v = iFact ? ((32 - iFact) * ref[b + iIdx + 1] + iFact * ref[b + iIdx + 2] + 16) >> 5 : ref[b + iIdx + 1];
      px_set(dst, pb, x + y * stride, v);
    }
  if (cIdx == 0 && nT < 32 && !disableBoundaryFilter) {
    if (mode == 26)
      for (int y = 0; y < nT; y++) px_set(dst, pb, y * stride, clip_bd(B(1) + ((B(-1 - y) - B(0)) >> 1), bit_depth));
    if (mode == 10)
      for (int x = 0; x < nT; x++) px_set(dst, pb, x, clip_bd(B(-1) + ((B(1 + x) - B(0)) >> 1), bit_depth));
  }
}

/* intrapred.h:185-258 */
void o_intra_sample_filtering(void* border, int pb, int nT, int cIdx, int mode, int strong, int bd_luma) {
  if (mode == 1 /*INTRA_DC*/ || nT == 4) return;
  const int minDist = imin(iabs(mode - 26), iabs(mode - 10));
  int filter;
  switch (nT) {
    case 8: filter = minDist > 7; break;
    case 16: filter = minDist > 1; break;
    case 32: filter = minDist > 0; break;
    default: filter = 0; break;
  }
  if (!filter) return;
  int pF[4 * 32 + 1];
  int* f = pF + 2 * 32;
  const int bi = strong && cIdx == 0 && nT == 32 && iabs(B(0) + B(64) - 2 * B(32)) < (1 << (bd_luma - 5)) &&
                 iabs(B(0) + B(-64) - 2 * B(-32)) < (1 << (bd_luma - 5));
  f[-2 * nT] = B(-2 * nT);
  f[2 * nT] = B(2 * nT);
  if (bi) {
    f[0] = B(0);
    for (int i = 1; i <= 63; i++) {
      f[-i] = B(0) + ((i * (B(-64) - B(0)) + 32) >> 6);
      f[i] = B(0) + ((i * (B(64) - B(0)) + 32) >> 6);
    }
  } else {
    for (int i = -(2 * nT - 1); i <= 2 * nT - 1; i++) f[i] = (B(i + 1) + 2 * B(i) + B(i - 1) + 2) >> 2;
  }
  for (int i = -2 * nT; i <= 2 * nT; i++) px_set(border, pb, i, f[i]);
}
#undef B

/* ------------------------------------------------------------------------ deblocking (A6) ----- */

void o_deblock_luma(void* ptr, ptrdiff_t stride, int pb, int vertical, int dE, int dEp, int dEq, int tc, int filterP,
                    int filterQ, int bd) {
  const ptrdiff_t across = vertical ? 1 : stride, along = vertical ? stride : 1;
  for (int k = 0; k < 4; k++) {
    const ptrdiff_t o = k * along;
    int p0 = px_get(ptr, pb, o - 1 * across), p1 = px_get(ptr, pb, o - 2 * across);
    int p2 = px_get(ptr, pb, o - 3 * across), p3 = px_get(ptr, pb, o - 4 * across);
    int q0 = px_get(ptr, pb, o), q1 = px_get(ptr, pb, o + across);
    int q2 = px_get(ptr, pb, o + 2 * across), q3 = px_get(ptr, pb, o + 3 * across);
    if (dE == 2) {
      int pn[3], qn[3];
      pn[0] = clip3(p0 - 2 * tc, p0 + 2 * tc, (p2 + 2 * p1 + 2 * p0 + 2 * q0 + q1 + 4) >> 3);
      pn[1] = clip3(p1 - 2 * tc, p1 + 2 * tc, (p2 + p1 + p0 + q0 + 2) >> 2);
      pn[2] = clip3(p2 - 2 * tc, p2 + 2 * tc, (2 * p3 + 3 * p2 + p1 + p0 + q0 + 4) >> 3);
      qn[0] = clip3(q0 - 2 * tc, q0 + 2 * tc, (p1 + 2 * p0 + 2 * q0 + 2 * q1 + q2 + 4) >> 3);
      qn[1] = clip3(q1 - 2 * tc, q1 + 2 * tc, (p0 + q0 + q1 + q2 + 2) >> 2);
      qn[2] = clip3(q2 - 2 * tc, q2 + 2 * tc, (p0 + q0 + q1 + 3 * q2 + 2 * q3 + 4) >> 3);
      for (int i = 0; i < 3; i++) {
        /* stored through pixel_t, no bit-depth clip (fallback-deblk.h:50-56) */
        if (filterP) px_set(ptr, pb, o - (i + 1) * across, pn[i]);
        if (filterQ) px_set(ptr, pb, o + i * across, qn[i]);
      }
    } else {
      int delta = (9 * (q0 - p0) - 3 * (q1 - p1) + 8) >> 4;
      if (iabs(delta) < tc * 10) {
        delta = clip3(-tc, tc, delta);
        if (filterP) px_set(ptr, pb, o - across, clip_bd(p0 + delta, bd));
        if (filterQ) px_set(ptr, pb, o, clip_bd(q0 - delta, bd));
        if (dEp == 1 && filterP) {
          int dp = clip3(-(tc >> 1), tc >> 1, (((p2 + p0 + 1) >> 1) - p1 + delta) >> 1);
          px_set(ptr, pb, o - 2 * across, clip_bd(p1 + dp, bd));
        }
        if (dEq == 1 && filterQ) {
          int dq = clip3(-(tc >> 1), tc >> 1, (((q2 + q0 + 1) >> 1) - q1 - delta) >> 1);
          px_set(ptr, pb, o + across, clip_bd(q1 + dq, bd));
        }
      }
    }
  }
}
void o_deblock_chroma(void* ptr, ptrdiff_t stride, int pb, int vertical, int tc, int filterP, int filterQ, int bd) {
  const ptrdiff_t across = vertical ? 1 : stride, along = vertical ? stride : 1;
  for (int k = 0; k < 4; k++) {
    const ptrdiff_t o = k * along;
    int p0 = px_get(ptr, pb, o - across), p1 = px_get(ptr, pb, o - 2 * across);
    int q0 = px_get(ptr, pb, o), q1 = px_get(ptr, pb, o + across);
    int delta = clip3(-tc, tc, ((((q0 - p0) * 4) + p1 - q1 + 4) >> 3));
    if (filterP) px_set(ptr, pb, o - across, clip_bd(p0 + delta, bd));
    if (filterQ) px_set(ptr, pb, o, clip_bd(q0 - delta, bd));
  }
}

/* ================================================================== picture level ============== */

o_frame* o_frame_new(int width, int height, int cf, int bdl, int bdc) {
  o_frame* f = (o_frame*)calloc(1, sizeof(o_frame));
  f->width = width; f->height = height; f->chroma_format = cf; f->bd_luma = bdl; f->bd_chroma = bdc;
  const int sw = (cf == 1 || cf == 2) ? 2 : 1, sh = (cf == 1) ? 2 : 1;
  for (int c = 0; c < 3; c++) {
    if (c > 0 && cf == 0) { f->w[c] = f->h[c] = 0; f->stride[c] = 0; f->p[c] = NULL; continue; }
    f->w[c] = c ? width / sw : width;
    f->h[c] = c ? height / sh : height;
    f->stride[c] = (f->w[c] + 15) & ~15; /* image.cc:113-117 */
    f->p[c] = (uint16_t*)calloc((size_t)f->stride[c] * f->h[c] + 64, sizeof(uint16_t)); /* zero: image.cc:164 */
  }
  return f;
}
void o_frame_free(o_frame* f) {
  if (!f) return;
  for (int c = 0; c < 3; c++) free(f->p[c]);
  free(f);
}
void o_frame_import(o_frame* f, int c, const void* src, ptrdiff_t stride, int bps) {
  for (int y = 0; y < f->h[c]; y++)
    for (int x = 0; x < f->w[c]; x++) f->p[c][y * f->stride[c] + x] = (uint16_t)px_get(src, bps, y * stride + x);
}
void o_frame_export(const o_frame* f, int c, void* dst, ptrdiff_t stride, int bps) {
  for (int y = 0; y < f->h[c]; y++)
    for (int x = 0; x < f->w[c]; x++) px_set(dst, bps, y * stride + x, f->p[c][y * f->stride[c] + x]);
}

/* rasterised per-picture metadata (what de265_image keeps in cb_info/pb_info/tu_info/deblk_info,
 * image.h:389-395) */
typedef struct pic_state {
  const m355_picture* pic;
  const m355_pic_params* pp;
  o_frame* dst;
  o_frame* const* refs;
  int sw, sh;                 /* SubWidthC, SubHeightC */
  int ctbW, ctbH, nCtb;       /* picture size in CTBs */
  int w4, h4;                 /* picture size in 4x4 units (deblk_info, image.cc:414-420) */
  int wcb, hcb;               /* in min-CB units */
  uint32_t* ctb_ts;           /* CtbAddrRStoTS */
  uint16_t* tile_id;          /* TileIdRS */
  uint32_t* cb_cu;            /* per min CB: index into cus[] (+1; 0 = none) */
  uint8_t* edge;              /* per 4x4: bit0 TU-edge V, bit1 TU-edge H, bit2 PB-edge V, bit3 PB-edge H, bit4 nonzero */
  uint32_t* pb_of;            /* per 4x4: index into pbs[] (+1; 0 = none) */
  uint8_t* bs;                /* per 4x4 boundary strength of the current pass */
  int16_t* resbuf;
} pic_state;

enum { E_TU_V = 1, E_TU_H = 2, E_PB_V = 4, E_PB_H = 8, E_NONZERO = 16 };

static int ctb_of(const pic_state* s, int xl, int yl) {
  return (yl >> s->pp->log2_ctb_size) * s->ctbW + (xl >> s->pp->log2_ctb_size);
}
static const m355_slice* slice_at(const pic_state* s, int xl, int yl) {
  return &s->pic->slices[s->pic->ctbs[ctb_of(s, xl, yl)].slice_idx];
}
static const m355_cu* cu_at(const pic_state* s, int xl, int yl) {
  uint32_t i = s->cb_cu[(yl >> s->pp->log2_min_cb_size) * s->wcb + (xl >> s->pp->log2_min_cb_size)];
  return i ? &s->pic->cus[i - 1] : NULL;
}
static int pred_mode_at(const pic_state* s, int xl, int yl) {
  const m355_cu* cu = cu_at(s, xl, yl);
  return cu ? cu->pred_mode : 0; /* cb_info zero-initialised: MODE_INTRA (image.h:173-195) */
}
static int qpy_at(const pic_state* s, int xl, int yl) { const m355_cu* cu = cu_at(s, xl, yl); return cu ? cu->qp_y : 0; }
static int pcm_at(const pic_state* s, int xl, int yl) { const m355_cu* cu = cu_at(s, xl, yl); return cu && (cu->flags & M355_CUF_PCM); }
static int bypass_at(const pic_state* s, int xl, int yl) { const m355_cu* cu = cu_at(s, xl, yl); return cu && (cu->flags & M355_CUF_TRANSQUANT_BYPASS); }

/* pps.cc:608-623 MinTbAddrZS */
static uint32_t min_tb_addr_zs(const pic_state* s, int xl, int yl) {
  const int shift = s->pp->log2_ctb_size - s->pp->log2_min_tb_size;
  const int x = xl >> s->pp->log2_min_tb_size, y = yl >> s->pp->log2_min_tb_size;
  uint32_t v = s->ctb_ts[ctb_of(s, xl, yl)] << (shift * 2);
  uint32_t p = 0;
  for (int i = 0; i < shift; i++) {
    int m = 1 << i;
    p += ((m & x) ? m * m : 0) + ((m & y) ? 2 * m * m : 0);
  }
  return v + p;
}

static int build_state(pic_state* s) {
  const m355_picture* pic = s->pic;
  const m355_pic_params* pp = s->pp;
  const int cs = 1 << pp->log2_ctb_size;
  s->sw = (pp->chroma_format_idc == 1 || pp->chroma_format_idc == 2) ? 2 : 1;
  s->sh = (pp->chroma_format_idc == 1) ? 2 : 1;
  s->ctbW = (pp->width + cs - 1) / cs;
  s->ctbH = (pp->height + cs - 1) / cs;
  s->nCtb = s->ctbW * s->ctbH;
  if (pic->n_ctbs != s->nCtb) return M355_ERR_INVALID;
  s->w4 = (pp->width + 3) / 4; s->h4 = (pp->height + 3) / 4;
  s->wcb = (pp->width + (1 << pp->log2_min_cb_size) - 1) >> pp->log2_min_cb_size;
  s->hcb = (pp->height + (1 << pp->log2_min_cb_size) - 1) >> pp->log2_min_cb_size;
  s->ctb_ts = (uint32_t*)calloc(s->nCtb, 4);
  s->tile_id = (uint16_t*)calloc(s->nCtb, 2);
  s->cb_cu = (uint32_t*)calloc((size_t)s->wcb * s->hcb, 4);
  s->edge = (uint8_t*)calloc((size_t)s->w4 * s->h4, 1);
  s->pb_of = (uint32_t*)calloc((size_t)s->w4 * s->h4, 4);
  s->bs = (uint8_t*)calloc((size_t)s->w4 * s->h4, 1);
  s->resbuf = (int16_t*)calloc((size_t)pic->res_len + 1, 2);
  /* 6.5.1 (pps.cc:589-606) */
  uint32_t ts = 0; int tidx = 0;
  for (int ty = 0; ty < pp->num_tile_rows; ty++)
    for (int tx = 0; tx < pp->num_tile_cols; tx++) {
      for (int y = pp->row_bd[ty]; y < pp->row_bd[ty + 1]; y++)
        for (int x = pp->col_bd[tx]; x < pp->col_bd[tx + 1]; x++) {
          if (x >= s->ctbW || y >= s->ctbH) return M355_ERR_INVALID;
          s->ctb_ts[y * s->ctbW + x] = ts++;
          s->tile_id[y * s->ctbW + x] = (uint16_t)tidx;
        }
      tidx++;
    }
  if ((int)ts != s->nCtb) return M355_ERR_INVALID;
  /* CU plane */
  for (int i = 0; i < pic->n_cus; i++) {
    const m355_cu* cu = &pic->cus[i];
    const int n = 1 << (cu->log2_size - pp->log2_min_cb_size);
    const int cx = cu->x >> pp->log2_min_cb_size, cy = cu->y >> pp->log2_min_cb_size;
    for (int y = cy; y < cy + n && y < s->hcb; y++)
      for (int x = cx; x < cx + n && x < s->wcb; x++) s->cb_cu[y * s->wcb + x] = (uint32_t)i + 1;
  }
  /* PB plane (pb_info) */
  for (int i = 0; i < pic->n_pbs; i++) {
    const m355_pb* pb = &pic->pbs[i];
    for (int y = pb->y / 4; y < (pb->y + pb->h) / 4 && y < s->h4; y++)
      for (int x = pb->x / 4; x < (pb->x + pb->w) / 4 && x < s->w4; x++) s->pb_of[y * s->w4 + x] = (uint32_t)i + 1;
  }
  return 0;
}
static void free_state(pic_state* s) {
  free(s->ctb_ts); free(s->tile_id); free(s->cb_cu); free(s->edge); free(s->pb_of); free(s->bs); free(s->resbuf);
}

/* ------------------------------------------------------------------------ inter (a8-a10) ------ */

/* mc_luma / mc_chroma (motion.cc:48-174, :178-282): out-of-picture samples are the nearest in-picture
 * sample (coordinate clamp), which is what the reference's padbuf / direct paths both produce. */
static void mc_block(const pic_state* s, const o_frame* ref, int c, int xInt, int yInt, int W, int H, int xf, int yf,
                     int16_t* out, int out_stride) {
  const int ntaps = c ? 4 : 8, before = c ? 1 : 3;
  const int bd = c ? s->pp->bit_depth_chroma : s->pp->bit_depth_luma;
  const int pw = W + ntaps, ph = H + ntaps;
  uint16_t* pad = (uint16_t*)malloc(sizeof(uint16_t) * (size_t)pw * ph);
  for (int y = 0; y < ph; y++)
    for (int x = 0; x < pw; x++) {
      int xa = clip3(0, ref->w[c] - 1, xInt + x - before), ya = clip3(0, ref->h[c] - 1, yInt + y - before);
      pad[y * pw + x] = ref->p[c][ya * ref->stride[c] + xa];
    }
  interp(out, out_stride, pad + before * pw + before, pw, 2, W, H, xf, yf, bd, ntaps, before);
  free(pad);
}

static void do_inter(pic_state* s) {
  const m355_picture* pic = s->pic;
  int16_t* pred[2];
  pred[0] = (int16_t*)malloc(sizeof(int16_t) * 64 * 64);
  pred[1] = (int16_t*)malloc(sizeof(int16_t) * 64 * 64);
  for (int i = 0; i < pic->n_pbs; i++) {
    const m355_pb* pb = &pic->pbs[i];
    const int nc = s->pp->chroma_format_idc ? 3 : 1;
    for (int c = 0; c < nc; c++) {
      const int bd = c ? s->pp->bit_depth_chroma : s->pp->bit_depth_luma;
      const int W = c ? pb->w / s->sw : pb->w, H = c ? pb->h / s->sh : pb->h;
      const int xP = c ? pb->x / s->sw : pb->x, yP = c ? pb->y / s->sh : pb->y;
      int used[2] = {0, 0};
      for (int l = 0; l < 2; l++) {
        if (!(pb->flags & (M355_PBF_MC_L0 << l))) continue;
        used[l] = 1;
        if (pb->flags & (M355_PBF_FILL_L0 << l)) { /* motion.cc:362-376 */
          for (int k = 0; k < W * H; k++) pred[l][(k / W) * 64 + k % W] = 1 << 13;
          continue;
        }
        const o_frame* ref = s->refs[pb->ref_slot[l]];
        int mvx = pb->mv[l][0], mvy = pb->mv[l][1];
        int xf, yf, xi, yi;
        if (c == 0) { xf = mvx & 3; yf = mvy & 3; xi = xP + (mvx >> 2); yi = yP + (mvy >> 2); }
        else { /* motion.cc:196-203 */
          mvx *= 2 / s->sw; mvy *= 2 / s->sh;
          xf = mvx & 7; yf = mvy & 7; xi = xP + (mvx >> 3); yi = yP + (mvy >> 3);
        }
        mc_block(s, ref, c, xi, yi, W, H, xf, yf, pred[l], 64);
      }
      uint16_t* d = s->dst->p[c] + (ptrdiff_t)yP * s->dst->stride[c] + xP;
      const ptrdiff_t ds = s->dst->stride[c];
      if (used[0] && used[1]) {
        if (pb->flags & M355_PBF_WEIGHTED) {
          const m355_wt* w0 = &pic->wts[pb->wt_idx[0]]; const m355_wt* w1 = &pic->wts[pb->wt_idx[1]];
          o_put_weighted_bipred(d, ds, 2, pred[0], pred[1], 64, W, H, w0->w[c], w0->o[c], w1->w[c], w1->o[c],
                                c ? w0->log2wd_chroma : w0->log2wd_luma, bd);
        } else o_put_weighted_pred_avg(d, ds, 2, pred[0], pred[1], 64, W, H, bd);
      } else if (used[0] || used[1]) {
        const int l = used[0] ? 0 : 1;
        if (pb->flags & M355_PBF_WEIGHTED) {
          const m355_wt* w = &pic->wts[pb->wt_idx[l]];
          o_put_weighted_pred(d, ds, 2, pred[l], 64, W, H, w->w[c], w->o[c], c ? w->log2wd_chroma : w->log2wd_luma, bd);
        } else o_put_unweighted_pred(d, ds, 2, pred[l], 64, W, H, bd);
      }
    }
  }
  free(pred[0]); free(pred[1]);
}

/* ------------------------------------------------------------------------ residual (a2-a7) ---- */

static const int level_scale[6] = {40, 45, 51, 57, 64, 72}; /* transform.cc:358 */

/* residual of one block before it is added / stored: dequant + transform / skip / bypass (transform.cc:361-607) */
static void rb_residual(pic_state* s, const m355_rb* rb, int32_t* r) {
  const m355_picture* pic = s->pic;
  const m355_pic_params* pp = s->pp;
  const int nT = 1 << rb->log2_size, bd = rb->cidx ? pp->bit_depth_chroma : pp->bit_depth_luma;
  int16_t coeff[32 * 32];
  memset(coeff, 0, sizeof(int16_t) * nT * nT);
  /* --- inverse quantisation (transform.cc:408-525) --- */
  for (int k = 0; k < rb->ncoeff; k++) {
    const uint32_t e = pic->coeffs[rb->coeff_ofs + k];
    const int pos = e & 0xFFFF;
    const int lvl = (int16_t)(e >> 16);
    if (pos >= nT * nT) continue;
    if (rb->kind == M355_RK_BYPASS || (rb->flags & M355_RBF_DEQUANTIZED)) { coeff[pos] = (int16_t)lvl; continue; }
    int bdShift = bd + rb->log2_size - 5;
    int64_t fact;
    if (!(pp->flags & M355_PF_SCALING_LIST)) { bdShift -= 4; fact = (int64_t)level_scale[rb->qp % 6] << (rb->qp / 6); }
    else {
      static const int sz_ofs[4] = {0, 6 * 16, 6 * 16 + 6 * 64, 6 * 16 + 6 * 64 + 6 * 256};
      const uint8_t* scl = pic->scaling_factors + sz_ofs[rb->log2_size - 2] + (rb->matrix_id & 7) * nT * nT;
      fact = (int64_t)(scl[pos] * level_scale[rb->qp % 6]) << (rb->qp / 6);
    }
    const int64_t offset = (int64_t)1 << (bdShift - 1);
    int64_t v = ((int64_t)lvl * fact + offset) >> bdShift;
    coeff[pos] = (int16_t)(v < -32768 ? -32768 : (v > 32767 ? 32767 : v));
  }
  /* --- transform / skip / bypass (transform.cc:404-447, :530-607) --- */
  if (rb->flags & M355_RBF_ROTATE) o_rotate_coefficients(coeff, nT);
  if (rb->kind == M355_RK_BYPASS) {
    if (rb->flags & M355_RBF_RDPCM_V) o_transform_bypass_rdpcm_v(r, coeff, nT);
    else if (rb->flags & M355_RBF_RDPCM_H) o_transform_bypass_rdpcm_h(r, coeff, nT);
    else o_transform_bypass(r, coeff, nT);
  } else if (rb->kind == M355_RK_SKIP) {
    const int bdShift2 = 20 - bd, tsShift = 5 + rb->log2_size; /* transform.cc:550-554 (no extended precision) */
    if (rb->flags & M355_RBF_RDPCM_V) o_rdpcm_v(r, coeff, nT, tsShift, bdShift2);
    else if (rb->flags & M355_RBF_RDPCM_H) o_rdpcm_h(r, coeff, nT, tsShift, bdShift2);
    else o_transform_skip_residual(r, coeff, nT, tsShift, bdShift2);
  } else {
    /* with cross_component_prediction_enabled_flag the reference goes through transform_idct_NxN + add_residual
       (transform.cc:611-616) with max_coeff_bits 15: the same arithmetic as the *_add slots */
    inv_transform(r, coeff, nT, rb->kind == M355_RK_DST, 20 - bd, -32768, 32767);
  }
}

static void do_residual(pic_state* s) {
  const m355_picture* pic = s->pic;
  const m355_pic_params* pp = s->pp;
  int nrb = pic->rb_count[0] + pic->rb_count[1] + pic->rb_count[2] + pic->rb_count[3];
  for (int i = 0; i < nrb; i++) {
    const m355_rb* rb = &pic->rbs[i];
    const int nT = 1 << rb->log2_size, bd = rb->cidx ? pp->bit_depth_chroma : pp->bit_depth_luma;
    int32_t r[32 * 32];
    rb_residual(s, rb, r);
    /* cross-component prediction (transform.cc:244-260, slice.cc:3721-3760): chroma residual += (ResScaleVal *
       ((rY << BitDepthC) >> BitDepthY)) >> 3 with the luma residual of the same transform unit.  The reference does
       both shifts on the value cast to uint32_t, i.e. the right shift is LOGICAL (a negative rY loses its sign):
       restated literally. */
    if ((pp->flags & M355_PF_CROSS_COMPONENT_PRED) && rb->cidx && ((rb->matrix_id >> 4) & 7)) {
      const int v = (rb->matrix_id >> 4) & 7;
      const int res_scale = (rb->matrix_id & 0x80) ? -(1 << (v - 1)) : (1 << (v - 1));
      const int back = (rb->matrix_id & 8) ? 2 : 1;
      if (i - back >= 0 && pic->rbs[i - back].cidx == 0 && pic->rbs[i - back].log2_size == rb->log2_size) {
        int32_t rl[32 * 32];
        rb_residual(s, &pic->rbs[i - back], rl);
        for (int k = 0; k < nT * nT; k++)
          r[k] += (res_scale * (int32_t)(((uint32_t)rl[k] << pp->bit_depth_chroma) >> pp->bit_depth_luma)) >> 3;
      }
    }
    if (rb->flags & M355_RBF_DEFERRED) {
      for (int k = 0; k < nT * nT; k++) s->resbuf[rb->res_ofs + k] = (int16_t)clip3(-32768, 32767, r[k]);
    } else {
      uint16_t* d = s->dst->p[rb->cidx] + (ptrdiff_t)rb->y * s->dst->stride[rb->cidx] + rb->x;
      add_res(d, s->dst->stride[rb->cidx], 2, r, nT, bd);
    }
  }
}

/* -------------------------------------------------------------------------- intra (a11-a12) --- */

/* intra_border_computer::preproc + fill_from_image + reference_sample_substitution
 * (intrapred.h:436-674), then filtering and prediction (intrapred.cc:277-317). */
static void do_intra_block(pic_state* s, const m355_ib* ib) {
  const m355_pic_params* pp = s->pp;
  const int c = ib->cidx, nT = 1 << ib->log2_size;
  const int SubW = c ? s->sw : 1, SubH = c ? s->sh : 1;
  const int bd = c ? pp->bit_depth_chroma : pp->bit_depth_luma;
  uint16_t* img = s->dst->p[c];
  const ptrdiff_t stride = s->dst->stride[c];
  const int xB = ib->x, yB = ib->y;
  uint16_t* dstp = img + (ptrdiff_t)yB * stride + xB;

  if (ib->flags & M355_IBF_PCM) { /* slice.cc:4211-4255: samples already shifted by the recorder */
    for (int y = 0; y < nT; y++)
      for (int x = 0; x < nT; x++) dstp[y * stride + x] = s->pic->pcm[ib->res_ofs + y * nT + x];
    return;
  }

  uint16_t border_mem[4 * 64 + 1];
  uint8_t avail_mem[4 * 64 + 1];
  uint16_t* border = border_mem + 2 * 64;
  uint8_t* avail = avail_mem + 2 * 64;
  memset(avail_mem, 0, sizeof(avail_mem));
  memset(border_mem, 0, sizeof(border_mem));

  int xBL = xB * SubW, yBL = yB * SubH;
  int availableLeft = 1, availableTop = 1, availableTopRight = 1, availableTopLeft = 1;
  if (xBL == 0) { availableLeft = 0; availableTopLeft = 0; }
  if (yBL == 0) { availableTop = 0; availableTopLeft = 0; availableTopRight = 0; }
  if (xBL + nT * SubW >= pp->width) availableTopRight = 0;

  const int l2c = pp->log2_ctb_size;
  const int xCurr = xBL >> l2c, yCurr = yBL >> l2c, xLeft = (xBL - 1) >> l2c, xRight = (xBL + nT * SubW) >> l2c,
            yTop = (yBL - 1) >> l2c;
#define SLICEADDR(cx, cy) (s->pic->slices[s->pic->ctbs[(cy) * s->ctbW + (cx)].slice_idx].slice_addr_rs)
#define TILEID(cx, cy) ((int)s->tile_id[(cy) * s->ctbW + (cx)])
  const int curS = SLICEADDR(xCurr, yCurr), curT = TILEID(xCurr, yCurr);
  if (availableLeft && (SLICEADDR(xLeft, yCurr) != curS || TILEID(xLeft, yCurr) != curT)) availableLeft = 0;
  if (availableTop && (SLICEADDR(xCurr, yTop) != curS || TILEID(xCurr, yTop) != curT)) availableTop = 0;
  if (availableTopLeft && (SLICEADDR(xLeft, yTop) != curS || TILEID(xLeft, yTop) != curT)) availableTopLeft = 0;
  if (availableTopRight && (SLICEADDR(xRight, yTop) != curS || TILEID(xRight, yTop) != curT)) availableTopRight = 0;
#undef SLICEADDR
#undef TILEID

  int nBottom = pp->height - yB * SubH;
  nBottom = (nBottom + SubH - 1) / SubH;
  if (nBottom > 2 * nT) nBottom = 2 * nT;
  int nRight = pp->width - xB * SubW;
  nRight = (nRight + SubW - 1) / SubW;
  if (nRight > 2 * nT) nRight = 2 * nT;

  int nAvail = 0;
  uint16_t firstValue = 0;
  const uint32_t curAddr = min_tb_addr_zs(s, xBL, yBL);
  const int cip = (pp->flags & M355_PF_CONSTRAINED_INTRA_PRED) != 0;

  for (int y = nBottom - 1; y >= 0; y -= 4)
    if (availableLeft) {
      const int xN = (xB - 1) * SubW, yN = (yB + y) * SubH;
      int av = min_tb_addr_zs(s, xN, yN) <= curAddr;
      if (cip && pred_mode_at(s, xN, yN) != 0) av = 0;
      if (av) {
        if (!nAvail) firstValue = img[xB - 1 + (yB + y) * stride];
        for (int i = 0; i < 4; i++) {
          avail[-y + i - 1] = 1;
          border[-y + i - 1] = img[xB - 1 + (yB + y - i) * stride];
        }
        nAvail += 4;
      }
    }
  if (availableTopLeft) {
    const int xN = (xB - 1) * SubW, yN = (yB - 1) * SubH;
    int av = min_tb_addr_zs(s, xN, yN) <= curAddr;
    if (cip && pred_mode_at(s, xN, yN) != 0) av = 0;
    if (av) {
      if (!nAvail) firstValue = img[xB - 1 + (yB - 1) * stride];
      border[0] = img[xB - 1 + (yB - 1) * stride];
      avail[0] = 1;
      nAvail++;
    }
  }
  for (int x = 0; x < nRight; x += 4) {
    const int ba = (x < nT) ? availableTop : availableTopRight;
    if (ba) {
      const int xN = (xB + x) * SubW, yN = (yB - 1) * SubH;
      int av = min_tb_addr_zs(s, xN, yN) <= curAddr;
      if (cip && pred_mode_at(s, xN, yN) != 0) av = 0;
      if (av) {
        if (!nAvail) firstValue = img[xB + x + (yB - 1) * stride];
        for (int i = 0; i < 4; i++) {
          border[x + i + 1] = img[xB + x + i + (yB - 1) * stride];
          avail[x + i + 1] = 1;
        }
        nAvail += 4;
      }
    }
  }
  /* reference_sample_substitution (intrapred.h:637-665) */
  if (nAvail != 4 * nT + 1) {
    if (nAvail == 0) {
      for (int i = -2 * nT; i <= 2 * nT; i++) border[i] = (uint16_t)(1 << (bd - 1));
    } else {
      if (!avail[-2 * nT]) border[-2 * nT] = firstValue;
      for (int i = -2 * nT + 1; i <= 2 * nT; i++)
        if (!avail[i]) border[i] = border[i - 1];
    }
  }
  /* filtering (intrapred.cc:289-293) */
  if (!(pp->flags & M355_PF_INTRA_SMOOTHING_DISABLED) && (c == 0 || pp->chroma_format_idc == 3))
    o_intra_sample_filtering(border, 2, nT, c, ib->mode, (pp->flags & M355_PF_STRONG_INTRA_SMOOTHING) != 0,
                             pp->bit_depth_luma);
  if (ib->mode == 0) o_intra_pred_planar(dstp, stride, 2, nT, c, border);
  else if (ib->mode == 1) o_intra_pred_dc(dstp, stride, 2, nT, c, border);
  else
    o_intra_pred_angular(dstp, stride, 2, bd, (ib->flags & M355_IBF_DISABLE_BOUNDARY_FILTER) != 0, ib->mode, nT, c,
                         border);
  if (ib->flags & M355_IBF_HAS_RESIDUAL) {
    const int16_t* r = s->resbuf + ib->res_ofs;
    for (int y = 0; y < nT; y++)
      for (int x = 0; x < nT; x++) dstp[y * stride + x] = (uint16_t)clip_bd(dstp[y * stride + x] + r[y * nT + x], bd);
  }
}

static void do_intra(pic_state* s) {
  /* decode order = tile scan over CTBs, list order inside a CTB */
  uint32_t* ts2rs = (uint32_t*)malloc(sizeof(uint32_t) * s->nCtb);
  for (int i = 0; i < s->nCtb; i++) ts2rs[s->ctb_ts[i]] = (uint32_t)i;
  for (int t = 0; t < s->nCtb; t++) {
    const m355_ctb* ctb = &s->pic->ctbs[ts2rs[t]];
    for (uint32_t k = 0; k < ctb->ib_count; k++) do_intra_block(s, &s->pic->ibs[ctb->ib_start + k]);
  }
  free(ts2rs);
}

/* -------------------------------------------------------------------------- deblock (a13-a15) - */

static const uint8_t tab_beta[52] = {0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  6,  7,
                                     8,  9,  10, 11, 12, 13, 14, 15, 16, 17, 18, 20, 22, 24, 26, 28, 30, 32,
                                     34, 36, 38, 40, 42, 44, 46, 48, 50, 52, 54, 56, 58, 60, 62, 64};
static const uint8_t tab_tc[54] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,  0,  0,  0,  0,
                                   1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 3,  3,  3,  3,  4,
                                   4, 4, 5, 5, 6, 6, 7, 8, 9, 10, 11, 13, 14, 16, 18, 20, 22, 24};
/* transform.h:29-34 table8_22 */
static int table8_22(int qPi) {
  static const int8_t t[14] = {29, 30, 31, 32, 33, 33, 34, 34, 35, 35, 36, 36, 37, 37};
  if (qPi < 30) return qPi;
  if (qPi >= 43) return qPi - 6;
  return t[qPi - 30];
}

/* derive_edgeFlags (deblock.cc:132-240) from the CU and TU-leaf lists */
static int derive_edge_flags(pic_state* s) {
  const m355_picture* pic = s->pic;
  const m355_pic_params* pp = s->pp;
  const int ctb_mask = (1 << pp->log2_ctb_size) - 1;
  int enabled = 0;
  /* per-CU: filterLeftCbEdge / filterTopCbEdge, stored for the TU pass */
  uint8_t* cuf = (uint8_t*)calloc(pic->n_cus + 1, 1); /* bit0 left ok, bit1 top ok, bit2 deblock enabled */
  for (int i = 0; i < pic->n_cus; i++) {
    const m355_cu* cu = &pic->cus[i];
    const int x0 = cu->x, y0 = cu->y;
    const m355_slice* sh = slice_at(s, x0, y0);
    int left = 1, top = 1;
    if (x0 == 0) left = 0;
    if (y0 == 0) top = 0;
    if (x0 && (x0 & ctb_mask) == 0) {
      if (!(sh->flags & M355_SF_LF_ACROSS_SLICES) && sh->slice_addr_rs != slice_at(s, x0 - 1, y0)->slice_addr_rs) left = 0;
      else if (!(pp->flags & M355_PF_LF_ACROSS_TILES) && s->tile_id[ctb_of(s, x0, y0)] != s->tile_id[ctb_of(s, x0 - 1, y0)]) left = 0;
    }
    if (y0 && (y0 & ctb_mask) == 0) {
      if (!(sh->flags & M355_SF_LF_ACROSS_SLICES) && sh->slice_addr_rs != slice_at(s, x0, y0 - 1)->slice_addr_rs) top = 0;
      else if (!(pp->flags & M355_PF_LF_ACROSS_TILES) && s->tile_id[ctb_of(s, x0, y0)] != s->tile_id[ctb_of(s, x0, y0 - 1)]) top = 0;
    }
    const int en = !(sh->flags & M355_SF_DEBLOCK_DISABLED);
    cuf[i] = (uint8_t)(left | (top << 1) | (en << 2));
    if (!en) continue;
    enabled = 1;
    /* markPredictionBlockBoundary (deblock.cc:68-129) */
    const int cb = 1 << cu->log2_size, h2 = cb >> 1, q4 = cb >> 2;
    int vx = -1, hy = -1;
    switch (cu->part_mode) {
      case 3: vx = h2; hy = h2; break;           /* NxN   */
      case 2: vx = h2; break;                    /* Nx2N  */
      case 1: hy = h2; break;                    /* 2NxN  */
      case 6: vx = q4; break;                    /* nLx2N */
      case 7: vx = h2 + q4; break;               /* nRx2N */
      case 4: hy = q4; break;                    /* 2NxnU */
      case 5: hy = h2 + q4; break;               /* 2NxnD */
      default: break;
    }
    for (int k = 0; k < cb; k++) {
      if (vx >= 0 && (y0 + k) / 4 < s->h4 && (x0 + vx) / 4 < s->w4) s->edge[((y0 + k) / 4) * s->w4 + (x0 + vx) / 4] |= E_PB_V;
      if (hy >= 0 && (y0 + hy) / 4 < s->h4 && (x0 + k) / 4 < s->w4) s->edge[((y0 + hy) / 4) * s->w4 + (x0 + k) / 4] |= E_PB_H;
    }
  }
  /* markTransformBlockBoundary (deblock.cc:33-63) on the leaves */
  for (int i = 0; i < pic->n_tus; i++) {
    const m355_tu* tu = &pic->tus[i];
    const int n = 1 << tu->log2_size;
    if (tu->flags & M355_TUF_NONZERO_COEFF)
      for (int y = tu->y / 4; y < (tu->y + n) / 4 && y < s->h4; y++)
        for (int x = tu->x / 4; x < (tu->x + n) / 4 && x < s->w4; x++) s->edge[y * s->w4 + x] |= E_NONZERO;
    uint32_t ci = s->cb_cu[(tu->y >> pp->log2_min_cb_size) * s->wcb + (tu->x >> pp->log2_min_cb_size)];
    if (!ci) continue;
    const m355_cu* cu = &pic->cus[ci - 1];
    const uint8_t f = cuf[ci - 1];
    if (!(f & 4)) continue;
    const int left = (tu->x == cu->x) ? (f & 1) : 1;
    const int top = (tu->y == cu->y) ? ((f >> 1) & 1) : 1;
    for (int k = 0; k < n; k += 4) {
      if (left && (tu->y + k) / 4 < s->h4) s->edge[((tu->y + k) / 4) * s->w4 + tu->x / 4] |= E_TU_V;
      if (top && (tu->x + k) / 4 < s->w4) s->edge[(tu->y / 4) * s->w4 + (tu->x + k) / 4] |= E_TU_H;
    }
  }
  free(cuf);
  return enabled;
}

/* derive_boundaryStrength (deblock.cc:243-383) */
static void derive_bs(pic_state* s, int vertical) {
  const m355_picture* pic = s->pic;
  const int xIncr = vertical ? 2 : 1, yIncr = vertical ? 1 : 2;
  const int xOffs = vertical ? 1 : 0, yOffs = vertical ? 0 : 1;
  const int edgeMask = vertical ? (E_TU_V | E_PB_V) : (E_TU_H | E_PB_H);
  const int tuMask = vertical ? E_TU_V : E_TU_H;
  for (int y = 0; y < s->h4; y += yIncr)
    for (int x = 0; x < s->w4; x += xIncr) {
      const int xDi = x << 2, yDi = y << 2;
      const uint8_t ef = s->edge[y * s->w4 + x];
      int bS = 0;
      if (ef & edgeMask) {
        const int xo = xDi - xOffs, yo = yDi - yOffs;
        if (pred_mode_at(s, xo, yo) == 0 || pred_mode_at(s, xDi, yDi) == 0) bS = 2;
        else if ((ef & tuMask) && ((ef & E_NONZERO) || (s->edge[(yo / 4) * s->w4 + xo / 4] & E_NONZERO))) bS = 1;
        else {
          const uint32_t ip = s->pb_of[(yo / 4) * s->w4 + xo / 4], iq = s->pb_of[y * s->w4 + x];
          /* both sides are inter here, so both have a PB record */
          if (ip && iq) {
            const m355_pb* P = &pic->pbs[ip - 1]; const m355_pb* Q = &pic->pbs[iq - 1];
            const int pf0 = (P->flags & M355_PBF_PRED_L0) != 0, pf1 = (P->flags & M355_PBF_PRED_L1) != 0;
            const int qf0 = (Q->flags & M355_PBF_PRED_L0) != 0, qf1 = (Q->flags & M355_PBF_PRED_L1) != 0;
            const int rP0 = pf0 ? P->ref_slot[0] : -1, rP1 = pf1 ? P->ref_slot[1] : -1;
            const int rQ0 = qf0 ? Q->ref_slot[0] : -1, rQ1 = qf1 ? Q->ref_slot[1] : -1;
            const int same = (rP0 == rQ0 && rP1 == rQ1) || (rP0 == rQ1 && rP1 == rQ0);
            if (!same) bS = 1;
            else {
              int p0x = pf0 ? P->mv[0][0] : 0, p0y = pf0 ? P->mv[0][1] : 0, p1x = pf1 ? P->mv[1][0] : 0, p1y = pf1 ? P->mv[1][1] : 0;
              int q0x = qf0 ? Q->mv[0][0] : 0, q0y = qf0 ? Q->mv[0][1] : 0, q1x = qf1 ? Q->mv[1][0] : 0, q1y = qf1 ? Q->mv[1][1] : 0;
#define FAR(ax, ay, bx, by) (iabs((ax) - (bx)) >= 4 || iabs((ay) - (by)) >= 4)
              if (rP0 != rP1) {
                if (rP0 == rQ0) { if (FAR(p0x, p0y, q0x, q0y) || FAR(p1x, p1y, q1x, q1y)) bS = 1; }
                else { if (FAR(p0x, p0y, q1x, q1y) || FAR(p1x, p1y, q0x, q0y)) bS = 1; }
              } else {
                if ((FAR(p0x, p0y, q0x, q0y) || FAR(p1x, p1y, q1x, q1y)) && (FAR(p0x, p0y, q1x, q1y) || FAR(p1x, p1y, q0x, q0y))) bS = 1;
              }
#undef FAR
            }
          }
        }
      }
      s->bs[y * s->w4 + x] = (uint8_t)bS;
    }
}

/* edge_filtering_luma_internal (deblock.cc:412-605) */
static void filter_luma(pic_state* s, int vertical) {
  const m355_pic_params* pp = s->pp;
  const int xIncr = vertical ? 2 : 1, yIncr = vertical ? 1 : 2;
  const ptrdiff_t stride = s->dst->stride[0];
  const int bd = pp->bit_depth_luma;
  for (int y = 0; y < s->h4; y += yIncr)
    for (int x = 0; x < s->w4; x += xIncr) {
      const int xDi = x << 2, yDi = y << 2;
      const int bS = s->bs[y * s->w4 + x];
      if (bS <= 0) continue;
      uint16_t* ptr = s->dst->p[0] + (ptrdiff_t)yDi * stride + xDi;
      int p[4][4], q[4][4];
      for (int k = 0; k < 4; k++)
        for (int i = 0; i < 4; i++) {
          if (vertical) { q[k][i] = ptr[i + k * stride]; p[k][i] = ptr[-i - 1 + k * stride]; }
          else { q[k][i] = ptr[k + i * stride]; p[k][i] = ptr[k - (i + 1) * stride]; }
        }
      const int QP_Q = qpy_at(s, xDi, yDi), QP_P = vertical ? qpy_at(s, xDi - 1, yDi) : qpy_at(s, xDi, yDi - 1);
      const int qP_L = (QP_Q + QP_P + 1) >> 1;
      const m355_slice* sh = slice_at(s, xDi, yDi);
      const int beta = tab_beta[clip3(0, 51, qP_L + sh->beta_offset)] * (1 << (bd - 8));
      const int tc = tab_tc[clip3(0, 53, qP_L + 2 * (bS - 1) + sh->tc_offset)] * (1 << (bd - 8));
      int dE = 0, dEp = 0, dEq = 0;
      const int dp0 = iabs(p[0][2] - 2 * p[0][1] + p[0][0]), dp3 = iabs(p[3][2] - 2 * p[3][1] + p[3][0]);
      const int dq0 = iabs(q[0][2] - 2 * q[0][1] + q[0][0]), dq3 = iabs(q[3][2] - 2 * q[3][1] + q[3][0]);
      const int dpq0 = dp0 + dq0, dpq3 = dp3 + dq3, dp = dp0 + dp3, dq = dq0 + dq3, d = dpq0 + dpq3;
      if (d < beta) {
        const int dSam0 = (2 * dpq0 < (beta >> 2) && iabs(p[0][3] - p[0][0]) + iabs(q[0][0] - q[0][3]) < (beta >> 3) &&
                           iabs(p[0][0] - q[0][0]) < ((5 * tc + 1) >> 1));
        const int dSam3 = (2 * dpq3 < (beta >> 2) && iabs(p[3][3] - p[3][0]) + iabs(q[3][0] - q[3][3]) < (beta >> 3) &&
                           iabs(p[3][0] - q[3][0]) < ((5 * tc + 1) >> 1));
        dE = (dSam0 && dSam3) ? 2 : 1;
        if (dp < ((beta + (beta >> 1)) >> 3)) dEp = 1;
        if (dq < ((beta + (beta >> 1)) >> 3)) dEq = 1;
      }
      if (dE) {
        const int xp = vertical ? xDi - 1 : xDi, yp = vertical ? yDi : yDi - 1;
        int filterP = 1, filterQ = 1;
        const int plf = (pp->flags & M355_PF_PCM_LOOP_FILTER_DISABLE) != 0;
        if (plf && pcm_at(s, xp, yp)) filterP = 0;
        if (bypass_at(s, xp, yp)) filterP = 0;
        if (plf && pcm_at(s, xDi, yDi)) filterQ = 0;
        if (bypass_at(s, xDi, yDi)) filterQ = 0;
        o_deblock_luma(ptr, stride, 2, vertical, dE, dEp, dEq, tc, filterP, filterQ, bd);
      }
    }
}

/* edge_filtering_chroma_internal (deblock.cc:635-761) */
static void filter_chroma(pic_state* s, int vertical) {
  const m355_pic_params* pp = s->pp;
  const int SW = s->sw, SH = s->sh;
  const int xIncr = (vertical ? 2 : 1) * SW, yIncr = (vertical ? 1 : 2) * SH;
  const ptrdiff_t stride = s->dst->stride[1];
  const int bd = pp->bit_depth_chroma;
  for (int y = 0; y < s->h4; y += yIncr)
    for (int x = 0; x < s->w4; x += xIncr) {
      const int xDi = x << (3 - SW), yDi = y << (3 - SH);
      const int xl = xDi * SW, yl = yDi * SH;
      const int bS = s->bs[(yl / 4) * s->w4 + xl / 4];
      if (bS <= 1) continue;
      for (int cp = 0; cp < 2; cp++) {
        const int cQpPicOffset = cp == 0 ? pp->pic_cb_qp_offset : pp->pic_cr_qp_offset;
        uint16_t* ptr = s->dst->p[cp + 1] + (ptrdiff_t)yDi * stride + xDi;
        const int QP_Q = qpy_at(s, xl, yl), QP_P = vertical ? qpy_at(s, xl - 1, yl) : qpy_at(s, xl, yl - 1);
        const int qP_i = ((QP_Q + QP_P + 1) >> 1) + cQpPicOffset;
        const int QP_C = (pp->chroma_format_idc == 1) ? table8_22(qP_i) : imin(qP_i, 51);
        const m355_slice* sh = slice_at(s, xl, yl);
        const int tc = tab_tc[clip3(0, 53, QP_C + 2 * (bS - 1) + sh->tc_offset)] * (1 << (bd - 8));
        const int xp = vertical ? xl - 1 : xl, yp = vertical ? yl : yl - 1;
        int filterP = 1, filterQ = 1;
        const int plf = (pp->flags & M355_PF_PCM_LOOP_FILTER_DISABLE) != 0;
        if (plf && pcm_at(s, xp, yp)) filterP = 0;
        if (bypass_at(s, xp, yp)) filterP = 0;
        if (plf && pcm_at(s, xl, yl)) filterQ = 0;
        if (bypass_at(s, xl, yl)) filterQ = 0;
        o_deblock_chroma(ptr, stride, 2, vertical, tc, filterP, filterQ, bd);
      }
    }
}

/* apply_deblocking_filter (deblock.cc:908-946) */
static void do_deblock(pic_state* s) {
  if (!derive_edge_flags(s)) return;
  for (int pass = 0; pass < 2; pass++) {
    const int vertical = pass == 0;
    derive_bs(s, vertical);
    filter_luma(s, vertical);
    if (s->pp->chroma_format_idc != 0) filter_chroma(s, vertical);
  }
}

/* ------------------------------------------------------------------------------- SAO (a16) ---- */

/* apply_sao_internal (sao.cc:28-263), reading `in` (the deblocked copy) and writing the picture */
static void sao_ctb(pic_state* s, int xCtb, int yCtb, int cIdx, const uint16_t* in, uint16_t* out, ptrdiff_t stride) {
  const m355_pic_params* pp = s->pp;
  const m355_ctb* ctb = &s->pic->ctbs[yCtb * s->ctbW + xCtb];
  const int type = (ctb->sao_type >> (2 * cIdx)) & 3;
  if (type == 0) return;
  const int bd = cIdx ? pp->bit_depth_chroma : pp->bit_depth_luma, maxv = (1 << bd) - 1;
  const int csw = cIdx ? (s->sw == 2) : 0, csh = cIdx ? (s->sh == 2) : 0; /* chroma shifts */
  const int nSW = (1 << pp->log2_ctb_size) >> csw, nSH = (1 << pp->log2_ctb_size) >> csh;
  const int xC = xCtb * nSW, yC = yCtb * nSH;
  const int width = s->dst->w[cIdx], height = s->dst->h[cIdx];
  /* NOTE (reference quirk, sao.cc:56): the CTB's slice address is looked up with the COMPONENT
     coordinates xC,yC used as if they were luma coordinates. */
  const int ctbSliceAddrRS = slice_at(s, imin(xC, pp->width - 1), imin(yC, pp->height - 1))->slice_addr_rs;
  const int ctbshiftW = pp->log2_ctb_size - csw, ctbshiftH = pp->log2_ctb_size - csh;
  const int ctbW = (xC + nSW > width) ? width - xC : nSW, ctbH = (yC + nSH > height) ? height - yC : nSH;
  const int extended = (ctb->flags & M355_CTBF_HAS_PCM_OR_BYPASS) != 0;
  const int plf = (pp->flags & M355_PF_PCM_LOOP_FILTER_DISABLE) != 0;
  if (type == 2) {
    int hPos[2], vPos[2];
    const int cls = (ctb->sao_eo_class >> (2 * cIdx)) & 3;
    switch (cls) {
      case 0: hPos[0] = -1; hPos[1] = 1; vPos[0] = 0; vPos[1] = 0; break;
      case 1: hPos[0] = 0; hPos[1] = 0; vPos[0] = -1; vPos[1] = 1; break;
      case 2: hPos[0] = -1; hPos[1] = 1; vPos[0] = -1; vPos[1] = 1; break;
      default: hPos[0] = 1; hPos[1] = -1; vPos[0] = -1; vPos[1] = 1; break;
    }
    int8_t off[5];
    off[0] = ctb->sao_offset[cIdx][0]; off[1] = ctb->sao_offset[cIdx][1]; off[2] = 0;
    off[3] = ctb->sao_offset[cIdx][2]; off[4] = ctb->sao_offset[cIdx][3];
    for (int j = 0; j < ctbH; j++)
      for (int i = 0; i < ctbW; i++) {
        const int xl = (xC + i) << csw, yl = (yC + j) << csh;
        if ((extended && plf && pcm_at(s, xl, yl)) || bypass_at(s, xl, yl)) continue;
        int edgeIdx = -1;
        if (i == 0 || j == 0 || i == ctbW - 1 || j == ctbH - 1)
          for (int k = 0; k < 2; k++) {
            const int xS = xC + i + hPos[k], yS = yC + j + vPos[k];
            if (xS < 0 || yS < 0 || xS >= width || yS >= height) { edgeIdx = 0; break; }
            const m355_slice* shN = slice_at(s, xS << csw, yS << csh);
            if (shN->slice_addr_rs < ctbSliceAddrRS && !(slice_at(s, xl, yl)->flags & M355_SF_LF_ACROSS_SLICES)) { edgeIdx = 0; break; }
            if (shN->slice_addr_rs > ctbSliceAddrRS && !(shN->flags & M355_SF_LF_ACROSS_SLICES)) { edgeIdx = 0; break; }
            if (!(pp->flags & M355_PF_LF_ACROSS_TILES) &&
                s->tile_id[(xS >> ctbshiftW) + (yS >> ctbshiftH) * s->ctbW] != s->tile_id[(xC >> ctbshiftW) + (yC >> ctbshiftH) * s->ctbW]) { edgeIdx = 0; break; }
          }
        if (edgeIdx != 0) {
          const ptrdiff_t o = (ptrdiff_t)(yC + j) * stride + xC + i;
          const int cur = in[o];
          edgeIdx = isign(cur - in[o + hPos[0] + vPos[0] * stride]) + isign(cur - in[o + hPos[1] + vPos[1] * stride]);
          out[o] = (uint16_t)clip3(0, maxv, cur + off[edgeIdx + 2]);
        }
      }
  } else {
    const int bandShift = bd - 5, left = ctb->sao_band_pos[cIdx];
    int bandTable[32];
    memset(bandTable, 0, sizeof(bandTable));
    for (int k = 0; k < 4; k++) bandTable[(k + left) & 31] = k + 1;
    for (int j = 0; j < ctbH; j++)
      for (int i = 0; i < ctbW; i++) {
        const int xl = (xC + i) << csw, yl = (yC + j) << csh;
        if (extended && ((plf && pcm_at(s, xl, yl)) || bypass_at(s, xl, yl))) continue;
        const ptrdiff_t o = (ptrdiff_t)(yC + j) * stride + xC + i;
        const int bandIdx = bandTable[clip3(0, maxv, in[o]) >> bandShift];
        if (bandIdx > 0) out[o] = (uint16_t)clip3(0, maxv, in[o] + ctb->sao_offset[cIdx][bandIdx - 1]);
      }
  }
}

/* apply_sample_adaptive_offset_sequential (sao.cc:327-382) */
static void do_sao(pic_state* s) {
  if (!(s->pp->flags & M355_PF_SAO_ENABLED)) return;
  const int nc = s->pp->chroma_format_idc ? 3 : 1;
  for (int c = 0; c < nc; c++) {
    const size_t n = (size_t)s->dst->stride[c] * s->dst->h[c];
    uint16_t* copy = (uint16_t*)malloc(n * 2);
    memcpy(copy, s->dst->p[c], n * 2);
    for (int y = 0; y < s->ctbH; y++)
      for (int x = 0; x < s->ctbW; x++) {
        const m355_slice* sh = &s->pic->slices[s->pic->ctbs[y * s->ctbW + x].slice_idx];
        if (c == 0 && !(sh->flags & M355_SF_SAO_LUMA)) continue;
        if (c != 0 && !(sh->flags & M355_SF_SAO_CHROMA)) continue;
        sao_ctb(s, x, y, c, copy, s->dst->p[c], s->dst->stride[c]);
      }
    free(copy);
  }
}

int o_decode_picture(const m355_picture* pic, o_frame* dst, o_frame* const* refs, int stages) {
  pic_state s;
  memset(&s, 0, sizeof(s));
  s.pic = pic; s.pp = &pic->pp; s.dst = dst; s.refs = refs;
  int err = build_state(&s);
  if (err) { free_state(&s); return err; }
  if (stages & M355_STAGE_INTER) do_inter(&s);
  if (stages & M355_STAGE_RESIDUAL) do_residual(&s);
  if (stages & M355_STAGE_INTRA) do_intra(&s);
  if ((stages & M355_STAGE_DEBLOCK) && (pic->pp.flags & M355_PF_DEBLOCK_ENABLED)) do_deblock(&s);
  if (stages & M355_STAGE_SAO) do_sao(&s);
  free_state(&s);
  return 0;
}

/* ================================================================================================
 * SEI decoded picture hash (sei.cc:161-257).  The hashed message is the plane row by row; samples of more than 8 bits
 * contribute two bytes, low byte first (raw_hash_data::prepare_16bit, sei.cc:141-158).
 * ================================================================================================ */

/* compute_checksum, sei.cc:161-186.  For bit_depth > 8 the reference indexes rows with stride/2 although its strides are
 * in samples (image.h:276-283), i.e. it reads the wrong rows; this restatement follows H.265 D.3.19 (and the reference's
 * 8-bit branch), which is what a stream's SEI carries. */
uint32_t o_hash_checksum(const void* data, int w, int h, ptrdiff_t stride, int bit_depth) {
  uint32_t sum = 0;
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      const uint8_t xorMask = (uint8_t)((x & 0xFF) ^ (y & 0xFF) ^ (x >> 8) ^ (y >> 8));
      if (bit_depth <= 8) sum += ((const uint8_t*)data)[y * stride + x] ^ xorMask;
      else {
        const uint16_t v = ((const uint16_t*)data)[y * stride + x];
        sum += (v & 0xFF) ^ xorMask;
        sum += (v >> 8) ^ xorMask;
      }
    }
  return sum;
}

/* crc_process_byte_parallel, sei.cc:198-207 */
static uint16_t crc_byte(uint16_t crc, uint8_t byte) {
  const uint16_t s = (uint16_t)(byte ^ (crc >> 8));
  const uint16_t t = (uint16_t)(s ^ (s >> 4));
  return (uint16_t)((crc << 8) ^ t ^ (t << 5) ^ (t << 12));
}
/* compute_CRC_8bit_fast, sei.cc:209-233 */
uint32_t o_hash_crc(const void* data, int w, int h, ptrdiff_t stride, int bit_depth) {
  uint16_t crc = 0xFFFF;
  crc = crc_byte(crc, 0);
  crc = crc_byte(crc, 0);
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      if (bit_depth <= 8) crc = crc_byte(crc, ((const uint8_t*)data)[y * stride + x]);
      else {
        const uint16_t v = ((const uint16_t*)data)[y * stride + x];
        crc = crc_byte(crc, (uint8_t)(v & 0xFF));
        crc = crc_byte(crc, (uint8_t)(v >> 8));
      }
    }
  return crc;
}

/* compute_MD5, sei.cc:236-257, over RFC 1321 (the reference bundles its own md5.cc) */
typedef struct { uint32_t s[4]; uint64_t n; uint8_t buf[64]; } o_md5;
static uint32_t md5_rol(uint32_t v, int r) { return (v << r) | (v >> (32 - r)); }
static void md5_block(o_md5* m, const uint8_t* p) {
  static const int R[4][4] = {{7, 12, 17, 22}, {5, 9, 14, 20}, {4, 11, 16, 23}, {6, 10, 15, 21}};
  uint32_t w[16], a = m->s[0], b = m->s[1], c = m->s[2], d = m->s[3];
  for (int i = 0; i < 16; i++) w[i] = (uint32_t)p[4 * i] | ((uint32_t)p[4 * i + 1] << 8) | ((uint32_t)p[4 * i + 2] << 16) | ((uint32_t)p[4 * i + 3] << 24);
  static uint32_t Kt[64];
  if (!Kt[0])   /* K[i] = floor(2^32 * |sin(i + 1)|) */
    for (int i = 63; i >= 0; i--) Kt[i] = (uint32_t)(int64_t)(4294967296.0 * __builtin_fabs(__builtin_sin((double)(i + 1))));
  for (int i = 0; i < 64; i++) {
    const uint32_t K = Kt[i];
    uint32_t f; int g;
    switch (i >> 4) {
      case 0: f = (b & c) | (~b & d); g = i; break;
      case 1: f = (d & b) | (~d & c); g = (5 * i + 1) & 15; break;
      case 2: f = b ^ c ^ d; g = (3 * i + 5) & 15; break;
      default: f = c ^ (b | ~d); g = (7 * i) & 15; break;
    }
    const uint32_t t = d;
    d = c; c = b;
    b = b + md5_rol(a + f + K + w[g], R[i >> 4][i & 3]);
    a = t;
  }
  m->s[0] += a; m->s[1] += b; m->s[2] += c; m->s[3] += d;
}
static void md5_update(o_md5* m, const uint8_t* p, size_t len) {
  for (size_t i = 0; i < len; i++) {
    m->buf[m->n++ & 63] = p[i];
    if ((m->n & 63) == 0) md5_block(m, m->buf);
  }
}
void o_hash_md5(const void* data, int w, int h, ptrdiff_t stride, int bit_depth, uint8_t out[16]) {
  o_md5 m = {{0x67452301u, 0xefcdab89u, 0x98badcfeu, 0x10325476u}, 0, {0}};
  const int bpp = bit_depth <= 8 ? 1 : 2;
  for (int y = 0; y < h; y++) md5_update(&m, (const uint8_t*)data + (size_t)y * stride * bpp, (size_t)w * bpp);   /* little-endian host = low byte first */
  const uint64_t bits = m.n * 8;
  const uint8_t pad = 0x80, zero = 0;
  md5_update(&m, &pad, 1);
  while ((m.n & 63) != 56) md5_update(&m, &zero, 1);
  uint8_t lenb[8];
  for (int i = 0; i < 8; i++) lenb[i] = (uint8_t)(bits >> (8 * i));
  md5_update(&m, lenb, 8);
  for (int i = 0; i < 4; i++)
    for (int k = 0; k < 4; k++) out[4 * i + k] = (uint8_t)(m.s[i] >> (8 * k));
}
