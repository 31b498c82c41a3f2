/*
 * k_residual.hip — inverse quantisation + inverse transform (DCT 4..32, DST 4x4, transform skip,
 * transquant bypass, RDPCM, rotation) for every coded transform block of a picture.
 *
 * Replaces scale_coefficients_internal (transform.cc:361-642) and the table slots it dispatches to:
 * dequant_coeff_block (fallback-dct.cc:1212-1220) + the int64 / scaling-list loops
 * (transform.cc:480-524), transform_idct_add / transform_4x4_luma_add (fallback-dct.cc:550-691,
 * 269-407), transform_skip_residual / rdpcm_* / transform_bypass* / rotate_coefficients
 * (fallback-dct.cc:81-256) and add_residual (fallback-dct.h:65-73).
 *
 * Mapping: blocks are binned by size on the host (m355_picture.rb_count); two launches (32/16 and 8/4, see
 * k_residual below), a wave handles 64/nT blocks of one size.  Within a wave, lane =
 * (column c, block b): the sparse (pos,level) pairs are dequantised and scattered into an LDS tile
 * stored as vertical int16 PAIRS, so the column pass is v_dot2c_i32_i16 over (row 2q, row 2q+1) with
 * the matrix pair coming from a compile-time table in constant memory — the matrix entry depends only
 * on (q, output row), not on the lane, so it is a SCALAR operand (s_load) and each LDS read feeds nT
 * dot2 issues.  The first-stage output is clipped to int16 exactly like the reference and written as
 * horizontal pairs; the row pass is the same scheme with lane = (row, block) and produces nT
 * horizontally adjacent samples per lane, which are added to the picture (inter) or stored to the
 * deferred-residual buffer (intra) with 16-byte vector accesses.  Both passes are pruned to the
 * occupied rows/columns (the reference's lastCol pruning, fallback-dct.cc:614-617, only skips zeros).
 * Roofline: HBM-bound on paper (4 B per nonzero coefficient in, nT^2 samples read-modify-write); in
 * practice VALU/latency-bound on dense blocks, see DESIGN.md.
 */
#include <stdlib.h>
#include <algorithm>
#include "k_common.h"
#include "k_meta_tu.h"
#include "k_intra_plan.h"


/* ---- compile-time tables: M[k][n] = c(k(2n+1)), c = quarter wave of the HEVC core transform
 * (the matrix of fallback-dct.cc:512-545), stored as int16 pairs (M[F*2q][i], M[F*(2q+1)][i]), F = 32/nT ---- */
struct ResTables {
  uint32_t dct[4][16 * 32];   /* [log2-2][q * nT + i] */
  uint32_t dst[2 * 4];        /* DST-VII 4x4 (fallback-dct.cc:260-265), same pairing */
};
constexpr int res_qw(int m)
{
  constexpr int qw[33] = {64, 90, 90, 90, 89, 88, 87, 85, 83, 82, 80, 78, 75, 73, 70, 67, 64,
                          61, 57, 54, 50, 46, 43, 38, 36, 31, 25, 22, 18, 13, 9,  4,  0};
  m &= 127;
  return m <= 32 ? qw[m] : (m <= 64 ? -qw[64 - m] : (m < 96 ? -qw[m - 64] : qw[128 - m]));
}
constexpr uint32_t res_pack(int lo, int hi) { return ((uint32_t)lo & 0xFFFFu) | ((uint32_t)hi << 16); }
constexpr ResTables make_res_tables()
{
  ResTables t{};
  for (int s = 0; s < 4; s++) {
    const int nT = 4 << s, F = 32 / nT;
    for (int q = 0; q < nT / 2; q++)
      for (int i = 0; i < nT; i++) t.dct[s][q * nT + i] = res_pack(res_qw((F * 2 * q) * (2 * i + 1)), res_qw((F * (2 * q + 1)) * (2 * i + 1)));
  }
  constexpr int d4[16] = {29, 55, 74, 84, 74, 74, 0, -74, 84, -29, -74, 55, 55, -84, 74, -29};
  for (int q = 0; q < 2; q++)
    for (int i = 0; i < 4; i++) t.dst[q * 4 + i] = res_pack(d4[(2 * q) * 4 + i], d4[(2 * q + 1) * 4 + i]);
  return t;
}
__constant__ ResTables c_res = make_res_tables();

__device__ __forceinline__ uint32_t d_sel_u(uint32_t mask, uint32_t a, uint32_t b) { return (a & mask) | (b & ~mask); }

/* waves per workgroup: the waves of a workgroup never cooperate (wave_sync only), so ONE — the finest scheduling grain and the
   smallest LDS footprint per workgroup (measured: within noise of 4 waves per workgroup, profiles/r02_b notes) */
#define RES_WPG 1
#define RES_LDS_DWORDS (2 * RES_WPG * (1024 + 32))   /* 32x32: 2 blocks per wave, nT^2/2 coefficient pairs + nT * (nT/2+1) first-stage pairs each */

/* Residual of one block per lane group (all lanes of the wave call this together): res[] = row `c` of the block,
 * NT adjacent samples, before it is added to the picture / stored.  smem pointers are the lane group's tile. */
#define RES_GB 8   /* (pos, level) pairs a lane fetches per batch */
template <int LOG2, bool PRE = false>
__device__ __forceinline__ void d_rb_compute(const DevPic& p, const m355_rb& rb, bool active, int c, uint32_t* cfp, int* res, const uint32_t* eb0 = nullptr)
{
  constexpr int NT = 1 << LOG2, N2 = NT * NT;
  constexpr int QN = NT / 2;             /* int16 pairs per column / row */
  constexpr int GP = QN + 1;             /* first-stage row pitch in dwords (padded: conflict-free row reads) */
  uint32_t* gp = cfp + QN * NT;                        /* gp[row * GP + q]  = (g[row][2q], g[row][2q+1])       */
  int16_t* cf16 = (int16_t*)cfp;
  int16_t* g16 = (int16_t*)gp;
  int* rr = (int*)cfp;                                 /* skip / bypass residual (int32, N2 <= BLK_DW) */
  const int bd = rb.cidx ? p.pp.bit_depth_chroma : p.pp.bit_depth_luma;

  /* ---- dequantise + scatter (transform.cc:408-525) ---- */
  wave_sync();                                         /* a previous use of the tile is over */
#pragma unroll
  for (int k = 0; k < QN; k++) cfp[k * NT + c] = 0;
  wave_sync();
  int maxrow = -1, maxcol = -1;
  {
    const bool raw = rb.kind == M355_RK_BYPASS || (rb.flags & M355_RBF_DEQUANTIZED);
    const bool sclist = (p.pp.flags & M355_PF_SCALING_LIST) != 0;
    int bdShift = bd + LOG2 - 5;
    if (!sclist) bdShift -= 4;
    const long long offset = 1ll << (bdShift - 1);
    /* levelScale[qP % 6] = {40, 45, 51, 57, 64, 72} (transform.cc:436) as a byte permute: the table in constant memory was a vector load per lane */
    const int ls = (int)d_byte_lookup(0x00004840u, 0x39332D28u, (unsigned)(rb.qp % 6)), qs = rb.qp / 6;
    const uint8_t* scl = nullptr;
    if (sclist && p.scaling) {
      const int sz_ofs = LOG2 == 2 ? 0 : LOG2 == 3 ? 6 * 16 : LOG2 == 4 ? 6 * 16 + 6 * 64 : 6 * 16 + 6 * 64 + 6 * 256;
      scl = p.scaling + sz_ofs + (rb.matrix_id & 7) * N2;
    }
    /* the (pos, level) pairs are fetched eight per lane at a time: a dense 32x32 block holds 1024 of them, and one
       dependent load -> scatter step per pair (32 memory round trips per lane) was what the whole launch waited for */
    constexpr int GB = RES_GB;
    auto scatter = [&](const uint32_t* eb) {
#pragma unroll
      for (int j = 0; j < GB; j++) {
        const uint32_t e = eb[j];
        int pos = e & 0xFFFF;
        const int lvl = (int16_t)(e >> 16);
        if (pos >= N2) continue;
        int v;
        if (raw) v = lvl;
        else {
          const long long fact = (long long)(scl ? scl[pos] * ls : ls) << qs;
          long long t = ((long long)lvl * fact + offset) >> bdShift;
          v = t < -32768 ? -32768 : (t > 32767 ? 32767 : (int)t);
        }
        if (rb.flags & M355_RBF_ROTATE) pos = N2 - 1 - pos;
        const int row = pos >> LOG2, col = pos & (NT - 1);
        cf16[((row >> 1) * NT + col) * 2 + (row & 1)] = (int16_t)v;
        if (v != 0) { maxrow = max(maxrow, row); maxcol = max(maxcol, col); }
      }
    };
    if (PRE) scatter(eb0);       /* the first batch was requested an iteration ago (d_res_issue) */
    for (int k0 = c + (PRE ? NT * GB : 0); k0 < rb.ncoeff; k0 += NT * GB) {
      uint32_t eb[GB];
#pragma unroll
      for (int j = 0; j < GB; j++) {
        const int k = k0 + j * NT;
        const uint32_t v = p.coeffs[rb.coeff_ofs + (uint32_t)min(k, max((int)rb.ncoeff - 1, 0))];   /* (clamped: no branch in front of a load) */
        eb[j] = k < rb.ncoeff ? v : 0xFFFFFFFFu;   /* pos 65535: skipped below */
      }
      scatter(eb);
    }
  }
  /* occupied extent: over the block's lanes for the skip/bypass test below, over the WAVE for the loop
     bounds (uniform bounds keep the matrix operands scalar; extra iterations only add zeros) */
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    maxrow = max(maxrow, __shfl_xor(maxrow, m, 64));
    maxcol = max(maxcol, __shfl_xor(maxcol, m, 64));
  }
  wave_sync();

  const bool is_tr = rb.kind == M355_RK_DCT || rb.kind == M355_RK_DST;
  const bool any_tr = __any(is_tr), any_other = __any(!is_tr && active);
#pragma unroll
  for (int i = 0; i < NT; i++) res[i] = 0;

  if (any_tr) {
    const uint32_t dstmask = ((LOG2 == 2) && rb.kind == M355_RK_DST) ? ~0u : 0u;
    const uint32_t* mt = c_res.dct[LOG2 - 2];
    /* column pass: g[i][col] = clip16((sum_j M[j][i] * coef[j][col] + 64) >> 7), lane = (col, block) */
    int acc[NT];
#pragma unroll
    for (int i = 0; i < NT; i++) acc[i] = 64;
    const int q1 = __builtin_amdgcn_readfirstlane((maxrow >> 1) + 1);   /* wave-uniform (tell hipcc: scalar loop, scalar matrix loads); maxrow = -1 -> 0 iterations */
    for (int q = 0; q < q1; q++) {
      const uint32_t v = cfp[q * NT + c];
#pragma unroll
      for (int i = 0; i < NT; i++) {
        uint32_t m = mt[q * NT + i];
        if (LOG2 == 2) { const uint32_t md = c_res.dst[q * 4 + i]; m = d_sel_u(dstmask, md, m); }   /* both scalar loads, per-lane select */
        acc[i] = d_dot2(v, m, acc[i]);
      }
    }
#pragma unroll
    for (int i = 0; i < NT; i++) g16[(i * GP) * 2 + c] = (int16_t)d_clip3(-32768, 32767, acc[i] >> 7);
    wave_sync();
    /* row pass: r[y][i] = (sum_j M[j][i] * g[y][j] + rnd) >> (20-bd), not clipped; lane = (row y, block) */
    const int postShift = 20 - bd;
#pragma unroll
    for (int i = 0; i < NT; i++) acc[i] = 1 << (postShift - 1);
    const int q2 = __builtin_amdgcn_readfirstlane((maxcol >> 1) + 1);
    for (int q = 0; q < q2; q++) {
      const uint32_t v = gp[c * GP + q];
#pragma unroll
      for (int i = 0; i < NT; i++) {
        uint32_t m = mt[q * NT + i];
        if (LOG2 == 2) { const uint32_t md = c_res.dst[q * 4 + i]; m = d_sel_u(dstmask, md, m); }   /* both scalar loads, per-lane select */
        acc[i] = d_dot2(v, m, acc[i]);
      }
    }
    if (is_tr) {
#pragma unroll
      for (int i = 0; i < NT; i++) res[i] = acc[i] >> postShift;
    }
    wave_sync();   /* gp/cfp are reused below (rr) only after everybody has read them */
  }
  if (any_other) {
    /* transform skip / bypass, optional RDPCM (fallback-dct.cc:81-91, 161-225); rare */
    const bool skip = rb.kind == M355_RK_SKIP;
    const int bdShift2 = 20 - bd, tsShift = 5 + LOG2, rnd = skip ? (1 << (bdShift2 - 1)) : 0;
    int col[NT];    /* column c of the block */
#pragma unroll
    for (int j = 0; j < NT; j++) {
      int v = cf16[((j >> 1) * NT + c) * 2 + (j & 1)];
      if (skip) v = ((int)((unsigned)v << tsShift) + rnd) >> bdShift2;
      col[j] = v;
    }
    wave_sync();   /* all reads of cf16 done before rr (same storage) is written */
    if (!is_tr) {
      if (rb.flags & M355_RBF_RDPCM_V) {
#pragma unroll
        for (int j = 1; j < NT; j++) col[j] += col[j - 1];
      }
#pragma unroll
      for (int j = 0; j < NT; j++) rr[j * NT + c] = col[j];
    }
    wave_sync();
    if (!is_tr) {
#pragma unroll
      for (int i = 0; i < NT; i++) res[i] = rr[c * NT + i];   /* row c */
      if (rb.flags & M355_RBF_RDPCM_H) {
#pragma unroll
        for (int i = 1; i < NT; i++) res[i] += res[i - 1];
      }
    }
  }
}

__device__ __forceinline__ m355_rb d_rb_idle(int log2)
{
  m355_rb rb;
  rb.ncoeff = 0; rb.kind = M355_RK_DCT; rb.flags = 0; rb.cidx = 0; rb.qp = 0; rb.x = rb.y = 0; rb.coeff_ofs = rb.res_ofs = 0; rb.matrix_id = 0; rb.log2_size = (uint8_t)log2;
  return rb;
}

/* One group = the 64 / nT blocks a wave handles together, in three steps so that the steps of consecutive groups can overlap
 * (tools/experiments/residual_sao_xcd_order_and_walk.patch): the record (d_res_record), the loads that depend on it (d_res_issue: the lane's destination row and — for the
 * loop — its first batch of coefficient pairs), and everything else (d_res_finish: scatter, transform, add / store). */
template <int LOG2, class PIX> struct ResGeom {
  static constexpr int NT = 1 << LOG2, BPW = 64 / NT, QN = NT / 2, GP = QN + 1, BLK_DW = QN * NT + NT * GP;
  static constexpr int NVP = sizeof(PIX) == 2 ? NT / 2 : (NT >= 4 ? NT / 4 : 1);     /* dwords of a destination row */
};

template <int LOG2>
__device__ __forceinline__ m355_rb d_res_record(const m355_rb* rbs, int rb_n, int tbi)
{
  m355_rb rb = d_rb_idle(LOG2);
  if (tbi < rb_n) rb = rbs[tbi];
  return rb;
}

template <int LOG2, class PIX, bool PRE>
__device__ __forceinline__ void d_res_issue(const DevPic& p, const m355_rb& rb, bool active, int c, int tbi, uint32_t* w, uint32_t* eb)
{
  constexpr int NT = ResGeom<LOG2, PIX>::NT, NVP = ResGeom<LOG2, PIX>::NVP;
  /* the lane's destination row (prediction samples the residual is added to) is requested BEFORE the transform: its
     memory round trip then overlaps the coefficient fetch and the two filter passes instead of following them (the
     launch is a chain of dependent round trips per workgroup, not bandwidth: 0.03-0.045 ms for any ONE block size alone) */
#pragma unroll
  for (int i = 0; i < NVP; i++) w[i] = 0;
  /* (an idle lane — no block — reads row 0: row c of a picture lower than the transform size does not exist) */
  const M355_GLOBAL PIX* d = (const M355_GLOBAL PIX*)M355_SEL3(p.plane, rb.cidx) + (size_t)(active ? rb.y + c : 0) * M355_SEL3(p.stride, rb.cidx) + rb.x;
  /* (requested by every lane, also where the row is not used — blocks of intra CUs, idle lanes; their address is a valid row of the
     plane all the same: a load under a per-lane condition leaves hipcc unsure of what is in flight behind it, and it then drains
     everything before the next load) */
  {
    if (sizeof(PIX) == 2) {
      if (NT >= 8) {
#pragma unroll
        for (int i = 0; i < NVP; i += 4) d_ldg16(d + 2 * i, w + i);
      } else d_ldg8(d, w);
    } else {
      if (NT >= 16) {
#pragma unroll
        for (int i = 0; i < NVP; i += 4) d_ldg16(d + 4 * i, w + i);
      } else if (NT == 8) d_ldg8(d, w);
      else w[0] = d_ldg4(d);
    }
  }
  if (PRE) {
#pragma unroll
    for (int j = 0; j < RES_GB; j++) {
      /* from a CLAMPED index, selected afterwards: a load under a per-lane condition is a branch and a wait of its own (eight
         dependent round trips instead of one) */
      const int k = c + j * NT;
      const uint32_t v = p.coeffs[rb.coeff_ofs + (uint32_t)min(k, max((int)rb.ncoeff - 1, 0))];
      eb[j] = k < rb.ncoeff ? v : 0xFFFFFFFFu;
    }
  }
}

/* a row of residuals added to the lane's destination row (w: read earlier) and written back: NT samples as 16-byte (NT >= 8) or 8-byte vectors — blocks are
   aligned to their size, so the row segment is too */
template <int LOG2, class PIX>
__device__ __forceinline__ void d_res_add_row(M355_GLOBAL PIX* d, uint32_t* w, const int* res, const int bd)
{
  constexpr int NT = 1 << LOG2;
  if (sizeof(PIX) == 2) {
    constexpr int NV = NT / 2;                 /* dwords per row */
#pragma unroll
    for (int i = 0; i < NV; i++)
      w[i] = (uint32_t)d_clip_bd((int)(w[i] & 0xFFFFu) + res[2 * i], bd) | ((uint32_t)d_clip_bd((int)(w[i] >> 16) + res[2 * i + 1], bd) << 16);
    if (NT >= 8) {
#pragma unroll
      for (int i = 0; i < NV; i += 4) d_stg16(d + 2 * i, w + i);
    } else d_stg8(d, w);
  } else {
    constexpr int NV = NT / 4;
#pragma unroll
    for (int i = 0; i < NV; i++)
      w[i] = (uint32_t)d_clip_bd((int)(w[i] & 0xFFu) + res[4 * i], bd) | ((uint32_t)d_clip_bd((int)((w[i] >> 8) & 0xFFu) + res[4 * i + 1], bd) << 8) |
             ((uint32_t)d_clip_bd((int)((w[i] >> 16) & 0xFFu) + res[4 * i + 2], bd) << 16) | ((uint32_t)d_clip_bd((int)(w[i] >> 24) + res[4 * i + 3], bd) << 24);
    if (NT >= 16) {
#pragma unroll
      for (int i = 0; i < NV; i += 4) d_stg16(d + 4 * i, w + i);
    } else if (NT == 8) d_stg8(d, w);
    else d_stg4(d, w[0]);
  }
}

template <int LOG2, class PIX, bool PRE>
__device__ __forceinline__ void d_res_finish(const DevPic& p, const m355_rb* rbs, const m355_rb& rb, bool active, int c, int tbi, uint32_t* cfp, uint32_t* w, const uint32_t* eb)
{
  constexpr int NT = ResGeom<LOG2, PIX>::NT;
  const int bd = rb.cidx ? p.pp.bit_depth_chroma : p.pp.bit_depth_luma;
  M355_GLOBAL PIX* d = (M355_GLOBAL PIX*)M355_SEL3(p.plane, rb.cidx) + (size_t)(rb.y + c) * M355_SEL3(p.stride, rb.cidx) + rb.x;

  int res[NT];      /* one row (lane's `c` is the row index here), NT adjacent samples */
  d_rb_compute<LOG2, PRE>(p, rb, active, c, cfp, res, eb);

  /* cross-component prediction (4:4:4 range extension; transform.cc:244-260, slice.cc:3721-3760): the chroma block
     adds (ResScaleVal * ((rY << BitDepthC) >> BitDepthY)) >> 3 of its transform unit's LUMA residual, which this lane
     group recomputes from the luma block's own coefficients (rbs[i-1] or rbs[i-2], see m355_rb.matrix_id) — no
     ordering between blocks, no side buffer.  Both shifts act on the value as uint32_t, as in the reference. */
  if (p.pp.flags & M355_PF_CROSS_COMPONENT_PRED) {
    const int v = (active && rb.cidx) ? ((rb.matrix_id >> 4) & 7) : 0;
    const int back = (rb.matrix_id & 8) ? 2 : 1;
    bool ccp = v != 0 && tbi - back >= 0;
    m355_rb rl = d_rb_idle(LOG2);
    if (ccp) {
      rl = rbs[tbi - back];
      if (rl.cidx != 0) { ccp = false; rl = d_rb_idle(LOG2); }
    }
    if (__any(ccp)) {
      int resl[NT];
      d_rb_compute<LOG2>(p, rl, ccp, c, cfp, resl);
      if (ccp) {
        const int scale = (rb.matrix_id & 0x80) ? -(1 << (v - 1)) : (1 << (v - 1));
#pragma unroll
        for (int i = 0; i < NT; i++)
          res[i] += (scale * (int)(((unsigned)resl[i] << p.pp.bit_depth_chroma) >> p.pp.bit_depth_luma)) >> 3;
      }
    }
  }

  if (!active) return;
  const int y = c;
  if ((rb.flags & M355_RBF_DEFERRED) || p.res_front) {
    /* (a chain picture's front launch: the block's tile — found by size bin and index, no offset table — instead of the add) */
    int16_t* out = (rb.flags & M355_RBF_DEFERRED) ? p.resbuf + rb.res_ofs + y * NT : p.res_tiles + p.res_tile_base[LOG2 - 2] + (size_t)tbi * (NT * NT) + y * NT;
#pragma unroll
    for (int i = 0; i < NT; i += 2) *(uint32_t*)(out + i) = res_pack(d_clip3(-32768, 32767, res[i]), d_clip3(-32768, 32767, res[i + 1]));
  } else d_res_add_row<LOG2, PIX>(d, w, res, bd);
}

template <int LOG2, class PIX>
__device__ __forceinline__ void d_residual_group(const DevPic& p, const m355_rb* rbs, int rb_n, int group, uint32_t* smem)
{
  typedef ResGeom<LOG2, PIX> G;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int c = lane & (G::NT - 1), b = lane >> LOG2;
  const int tbi = (group * RES_WPG + wave) * G::BPW + b;
  const bool active = tbi < rb_n;
  uint32_t* cfp = smem + (wave * G::BPW + b) * G::BLK_DW;   /* cfp[q * NT + col] = (coef[2q][col], coef[2q+1][col]) */
  const m355_rb rb = d_res_record<LOG2>(rbs, rb_n, tbi);
  /* the destination row and the first batch of coefficient pairs in ONE straight line of loads behind the record (a block of up to
     8 nT pairs has no other batch): requested inside the batch loop, the pairs waited for the row first — hipcc drains every load in
     flight at a loop header */
  uint32_t w[G::NVP], eb[RES_GB];
  d_res_issue<LOG2, PIX, true>(p, rb, active, c, tbi, w, eb);
  d_res_finish<LOG2, PIX, true>(p, rbs, rb, active, c, tbi, cfp, w, eb);
}

/* Two launches, issued side by side on the lane's two streams (runtime_decode.hip, launch_prediction): 32x32 + 16x16 blocks (2 / 4 per
 * wave; 128 VGPRs, 8 KB of LDS tiles per wave) and 8x8 + 4x4 blocks (8 / 16 per wave; 66 VGPRs -> 7 waves per SIMD instead of the
 * 3 the 32-point transform's registers would impose on every size).  The stage is a chain of dependent round trips per wave —
 * record, coefficient pairs, destination rows — so resident waves are what hides it.  Larger size first within each launch. */
#define RES_LDS_DWORDS_SMALL (8 * RES_WPG * (4 * 8 + 8 * 5))   /* 8x8: 8 blocks per wave (4x4: 16 x 20 dwords fit too) */
/* blocks per workgroup: RES_WPG waves * 64/nT */
__host__ __device__ static inline int res_groups(int n, int per_wave) { return (n + per_wave * RES_WPG - 1) / (per_wave * RES_WPG); }
template <class PIX, bool BIG>
__device__ __forceinline__ void k_residual_body(const DevPic& p, int ng_hi, uint32_t* s_buf, int g = (int)blockIdx.x, const bool xcd_order = false)
{
  M355_GATE(p);
  if (xcd_order)
  /* (the two-launch form: pictures on two-stream lanes) block b runs on XCD b % 8; each of the launch's two size bins is dealt to the XCDs as eight CONTIGUOUS runs of
     groups (= compact regions of the picture in decode order), both bins padded to a multiple of eight workgroups, so that the 128-byte lines a group's 8- / 16- / 32-byte
     rows lie in are fetched by ONE L2 — with the round-robin order the four groups of 4x4 blocks that share a line sit on four XCDs and each fetches it: a third less fabric
     fetch for the 8x8 + 4x4 launch, the stage 0.081 -> 0.076 ms at C5 (profiles/r04_af_*, r06_v52_*) */
  {
    const int n_hi = BIG ? res_groups(p.rb_count[3], 2) : res_groups(p.rb_count[1], 8), n_lo = BIG ? res_groups(p.rb_count[2], 4) : res_groups(p.rb_count[0], 16);
    const int n_hi8 = (n_hi + 7) & ~7;
    const bool hi = g < n_hi8;
    const int b = hi ? g : g - n_hi8, per = (hi ? n_hi8 : ((n_lo + 7) & ~7)) >> 3;
    const int gg = (b & 7) * per + (b >> 3);
    if (gg >= (hi ? n_hi : n_lo)) return;
    g = hi ? gg : ng_hi + gg;
  }
  if (BIG) {
    if (g < ng_hi) d_residual_group<5, PIX>(p, p.rb_bin[3], p.rb_count[3], g, s_buf);
    else d_residual_group<4, PIX>(p, p.rb_bin[2], p.rb_count[2], g - ng_hi, s_buf);
  } else {
    if (g < ng_hi) d_residual_group<3, PIX>(p, p.rb_bin[1], p.rb_count[1], g, s_buf);
    else d_residual_group<2, PIX>(p, p.rb_bin[0], p.rb_count[0], g - ng_hi, s_buf);
  }
}

template <class PIX, bool BIG>
__global__ void __launch_bounds__(64 * RES_WPG) __attribute__((amdgpu_waves_per_eu(4))) k_residual(DevPic p, int ng_hi)
{
  __shared__ __attribute__((aligned(16))) uint32_t s_buf[BIG ? RES_LDS_DWORDS : RES_LDS_DWORDS_SMALL];
  k_residual_body<PIX, BIG>(p, ng_hi, s_buf, (int)blockIdx.x, true);
}
/* BOTH launches as roles of one: workgroups [0, n_big) take the 32x32 + 16x16 groups, the rest the 8x8 + 4x4 groups.  For pictures on a one-stream lane
   (up to 4K: runtime_decode.hip launch_prediction), where the two launches stand one behind the other and neither fills the GPU — the occupancy the
   separate small-block kernel buys (66 registers) is not what a 4K picture's 13 us launch waits for */
template <class PIX>
__global__ void __launch_bounds__(64 * RES_WPG) __attribute__((amdgpu_waves_per_eu(4))) k_residual_both(DevPic p, int n_big, int ng_hi_big, int ng_hi_small)
{
  __shared__ __attribute__((aligned(16))) uint32_t s_buf[RES_LDS_DWORDS > RES_LDS_DWORDS_SMALL ? RES_LDS_DWORDS : RES_LDS_DWORDS_SMALL];
  const int g = (int)blockIdx.x;
  if (g < n_big) k_residual_body<PIX, true>(p, ng_hi_big, s_buf, g);
  else k_residual_body<PIX, false>(p, ng_hi_small, s_buf, g - n_big);
}
/* ... and with them the transform-edge scatter (k_meta_tu.h) and the border plans (k_intra_plan.h), which only k_intra and the deblocking filter read: workgroups
   [n_res, n_res + nb_tu) walk the transform leaves (64 per workgroup), the rest plan work item r / n_parts, part r % n_parts (one wave each).  On a one-stream
   lane they stood as a launch of their own (k_tu_plan) in front of k_inter: 14 us of a 4K picture's 176 */
template <class PIX, int CF>
__global__ void __launch_bounds__(64 * RES_WPG) __attribute__((amdgpu_waves_per_eu(4))) k_residual_tu_plan(DevPic p, int n_res, int n_big, int ng_hi_big, int ng_hi_small, int nb_tu, int work_n, int n_parts)
{
  __shared__ __attribute__((aligned(16))) uint32_t s_buf[RES_LDS_DWORDS > RES_LDS_DWORDS_SMALL ? RES_LDS_DWORDS : RES_LDS_DWORDS_SMALL];
  const int g = (int)blockIdx.x;
  if (g < n_big) { k_residual_body<PIX, true>(p, ng_hi_big, s_buf, g); return; }
  if (g < n_res) { k_residual_body<PIX, false>(p, ng_hi_small, s_buf, g - n_big); return; }
  const int q = g - n_res;
  if (q < nb_tu) { k_meta_tu_body(p, q); return; }
  k_intra_plan_body<CF>(p, work_n, (q - nb_tu) / n_parts, (q - nb_tu) % n_parts, n_parts);
}
/* k_residual_add: the tiles a res_front launch stored, added to the prediction samples (the add of d_res_finish, with the transform already done): one lane per
   block row, 64 / nT blocks per wave, every size bin a range of the grid, largest first.  One round trip for the record, one for the tile row and the destination row. */
template <int LOG2, class PIX>
__device__ __forceinline__ void d_residual_add_group(const DevPic& p, const m355_rb* rbs, int rb_n, int group)
{
  constexpr int NT = 1 << LOG2, BPW = 64 / NT;
  const int lane = threadIdx.x & 63, c = lane & (NT - 1), tbi = group * BPW + (lane >> LOG2);
  if (tbi >= rb_n) return;
  const m355_rb rb = rbs[tbi];
  if (rb.flags & M355_RBF_DEFERRED) return;                  /* (a block of an intra CU: k_intra adds its residual) */
  const int bd = rb.cidx ? p.pp.bit_depth_chroma : p.pp.bit_depth_luma;
  M355_GLOBAL PIX* d = (M355_GLOBAL PIX*)M355_SEL3(p.plane, rb.cidx) + (size_t)(rb.y + c) * M355_SEL3(p.stride, rb.cidx) + rb.x;
  const int16_t* t = p.res_tiles + p.res_tile_base[LOG2 - 2] + (size_t)tbi * (NT * NT) + c * NT;
  uint32_t rr[NT / 2], w[ResGeom<LOG2, PIX>::NVP];
  /* the tile row (2 NT bytes) and the destination row, both requested before either is used */
  if (NT >= 8) {
#pragma unroll
    for (int i = 0; i < NT / 2; i += 4) d_ldg16((const M355_GLOBAL int16_t*)t + 2 * i, rr + i);
  } else d_ldg8((const M355_GLOBAL int16_t*)t, rr);
  if (sizeof(PIX) == 2) {
    if (NT >= 8) {
#pragma unroll
      for (int i = 0; i < NT / 2; i += 4) d_ldg16(d + 2 * i, w + i);
    } else d_ldg8(d, w);
  } else {
    if (NT >= 16) {
#pragma unroll
      for (int i = 0; i < NT / 4; i += 4) d_ldg16(d + 4 * i, w + i);
    } else if (NT == 8) d_ldg8(d, w);
    else w[0] = d_ldg4(d);
  }
  int res[NT];
#pragma unroll
  for (int i = 0; i < NT; i++) res[i] = (int)(int16_t)(rr[i >> 1] >> (16 * (i & 1)));
  d_res_add_row<LOG2, PIX>(d, w, res, bd);
}
template <class PIX>
__global__ void __launch_bounds__(64) k_residual_add(DevPic p, int n5, int n4, int n3)
{
  M355_GATE(p);
  const int g = (int)blockIdx.x;
  if (g < n5) d_residual_add_group<5, PIX>(p, p.rb_bin[3], p.rb_count[3], g);
  else if (g < n5 + n4) d_residual_add_group<4, PIX>(p, p.rb_bin[2], p.rb_count[2], g - n5);
  else if (g < n5 + n4 + n3) d_residual_add_group<3, PIX>(p, p.rb_bin[1], p.rb_count[1], g - n5 - n4);
  else d_residual_add_group<2, PIX>(p, p.rb_bin[0], p.rb_count[0], g - n5 - n4 - n3);
}
void m355_launch_residual_add(const DevPic& p, bool hbd, hipStream_t st)
{
  const int n5 = (p.rb_count[3] + 1) / 2, n4 = (p.rb_count[2] + 3) / 4, n3 = (p.rb_count[1] + 7) / 8, n2 = (p.rb_count[0] + 15) / 16;
  if (!(n5 + n4 + n3 + n2)) return;
  if (hbd) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_residual_add<uint16_t>), dim3(n5 + n4 + n3 + n2), dim3(64), 0, st, p, n5, n4, n3);
  else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_residual_add<uint8_t>), dim3(n5 + n4 + n3 + n2), dim3(64), 0, st, p, n5, n4, n3);
}

/* batch form (intra pictures) */
template <class PIX, bool BIG>
__global__ void __launch_bounds__(64 * RES_WPG) __attribute__((amdgpu_waves_per_eu(4))) k_residual_batch(DevBatch b)
{
  __shared__ __attribute__((aligned(16))) uint32_t s_buf[BIG ? RES_LDS_DWORDS : RES_LDS_DWORDS_SMALL];
  M355_BATCH_PIC(b);
  k_residual_body<PIX, BIG>(p, BIG ? res_groups(p.rb_count[3], 2) : res_groups(p.rb_count[1], 8), s_buf);
}

template <class PIX, bool BIG>
static void launch_res_batch(const HostBatch& b, hipStream_t st)
{
  int n = 0;
  for (int k = 0; k < b.n; k++) if ((b.on >> k) & 1u) {
    const DevPic& p = b.host[k];
    n = std::max(n, BIG ? res_groups(p.rb_count[3], 2) + res_groups(p.rb_count[2], 4) : res_groups(p.rb_count[1], 8) + res_groups(p.rb_count[0], 16));
  }
  if (n) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_residual_batch<PIX, BIG>), dim3(n, 1, b.n), dim3(64 * RES_WPG), 0, st, DevBatch{b.dev, b.on});
}
void m355_launch_residual_batch(const HostBatch& b, bool hbd, bool big, hipStream_t st)
{
  if (big) { if (hbd) launch_res_batch<uint16_t, true>(b, st); else launch_res_batch<uint8_t, true>(b, st); }
  else { if (hbd) launch_res_batch<uint16_t, false>(b, st); else launch_res_batch<uint8_t, false>(b, st); }
}

template <class PIX, bool BIG>
static void launch_res(const DevPic& p, int n, int ng_hi, hipStream_t st)
{
  const dim3 blk(64 * RES_WPG);
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k_residual<PIX, BIG>), dim3(n), blk, 0, st, p, ng_hi);
}

void m355_launch_residual_both(const DevPic& p, bool hbd, hipStream_t st)
{
  const int ng2 = res_groups(p.rb_count[0], 16), ng3 = res_groups(p.rb_count[1], 8), ng4 = res_groups(p.rb_count[2], 4), ng5 = res_groups(p.rb_count[3], 2);
  const int n = ng5 + ng4 + ng3 + ng2;
  if (!n) return;
  if (hbd) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_residual_both<uint16_t>), dim3(n), dim3(64 * RES_WPG), 0, st, p, ng5 + ng4, ng5, ng3);
  else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_residual_both<uint8_t>), dim3(n), dim3(64 * RES_WPG), 0, st, p, ng5 + ng4, ng5, ng3);
}

template <class PIX>
static void launch_res_tu_plan(const DevPic& p, int n_res, int n_big, int ng5, int ng3, int nb_tu, int n_plan, int n_parts, hipStream_t st)
{
  const dim3 grid(n_res + nb_tu + n_plan * n_parts), blk(64 * RES_WPG);
  switch (p.pp.chroma_format_idc) {
    case 0: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_residual_tu_plan<PIX, 0>), grid, blk, 0, st, p, n_res, n_big, ng5, ng3, nb_tu, n_plan, n_parts); break;
    case 1: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_residual_tu_plan<PIX, 1>), grid, blk, 0, st, p, n_res, n_big, ng5, ng3, nb_tu, n_plan, n_parts); break;
    case 2: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_residual_tu_plan<PIX, 2>), grid, blk, 0, st, p, n_res, n_big, ng5, ng3, nb_tu, n_plan, n_parts); break;
    default: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_residual_tu_plan<PIX, 3>), grid, blk, 0, st, p, n_res, n_big, ng5, ng3, nb_tu, n_plan, n_parts); break;
  }
}
/* residuals (both size classes) + transform edges + border plans: the work m355_launch_residual_both and m355_launch_tu_plan (k_intra.hip) do, as one launch */
void m355_launch_residual_tu_plan(const DevPic& p, bool hbd, hipStream_t st)
{
  const int ng2 = res_groups(p.rb_count[0], 16), ng3 = res_groups(p.rb_count[1], 8), ng4 = res_groups(p.rb_count[2], 4), ng5 = res_groups(p.rb_count[3], 2);
  const int n_res = ng5 + ng4 + ng3 + ng2;
  /* (one intra picture at a time: its CTBs are planned by k_intra itself; an intra picture's CTB is planned by PLAN_SPLIT x 4 waves, an inter picture's by 4) */
  const int n_plan = (p.intra_dense && p.intra_keeper) ? 0 : p.n_intra_work, n_parts = 4 * (p.intra_dense ? PLAN_SPLIT : 1);
  const int nb_tu = (p.n_tus + 64 * RES_WPG - 1) / (64 * RES_WPG);
  if (!(n_res + nb_tu + n_plan)) return;
  if (hbd) launch_res_tu_plan<uint16_t>(p, n_res, ng5 + ng4, ng5, ng3, nb_tu, n_plan, n_parts, st);
  else launch_res_tu_plan<uint8_t>(p, n_res, ng5 + ng4, ng5, ng3, nb_tu, n_plan, n_parts, st);
}

void m355_launch_residual(const DevPic& p, bool hbd, bool big, hipStream_t st)
{
  auto groups = res_groups;
  const int ng2 = groups(p.rb_count[0], 16), ng3 = groups(p.rb_count[1], 8), ng4 = groups(p.rb_count[2], 4), ng5 = groups(p.rb_count[3], 2);
  auto pad8 = [](int n) { return (n + 7) & ~7; };      /* (k_residual_body: each bin as eight runs) */
  if (big && ng5 + ng4) {
    if (hbd) launch_res<uint16_t, true>(p, pad8(ng5) + pad8(ng4), ng5, st);
    else launch_res<uint8_t, true>(p, pad8(ng5) + pad8(ng4), ng5, st);
  }
  if (!big && ng3 + ng2) {
    if (hbd) launch_res<uint16_t, false>(p, pad8(ng3) + pad8(ng2), ng3, st);
    else launch_res<uint8_t, false>(p, pad8(ng3) + pad8(ng2), ng3, st);
  }
}
