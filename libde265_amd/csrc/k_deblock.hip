/*
 * k_deblock.hip — in-loop deblocking filter, one launch per edge direction.
 *
 * Replaces apply_deblocking_filter (deblock.cc:908-946): derive_boundaryStrength (:243-383),
 * edge_filtering_luma_internal (:412-605) with deblock_luma_kernel (fallback-deblk.h:33-100) and
 * edge_filtering_chroma_internal (:635-761) with deblock_chroma_kernel (fallback-deblk.h:104-124).
 * (Edge flags come from k_meta.hip.)  As in the reference, ALL vertical edges of the picture are
 * filtered before any horizontal edge — here a kernel boundary instead of the progress locks of
 * deblock.cc:826-844.  Within one direction every 8x8-grid edge segment touches disjoint samples
 * (<= 3 modified, 4 read, on each side of edges 8 apart), so the pass is embarrassingly parallel:
 * one thread per 4-line edge segment, which derives bS, beta/tc and the filter decisions and filters
 * the luma segment plus (every second segment, 4:2:0) the two chroma segments.
 * Roofline: HBM-bound — picture read + write once per direction plus ~1.3 B of metadata per 4x4.
 */
#include <algorithm>
#include "k_common.h"
#define K_DEBLOCK_DEV_OWNER
#include "k_deblock_dev.h"

/* where the samples of the three planes are: the picture itself (the two-pass kernels), or a workgroup's tile of it in LDS (k_deblock_tiles):
   sample (x, y) of plane c at pl[c] + (y - oy[c]) * st[c] + (x - ox[c]) */
template <class PIX> struct DbView { PIX* pl[3]; int st[3], ox[3], oy[3]; };
template <class PIX> __device__ __forceinline__ DbView<PIX> d_db_view_picture(const DevPic& p)
{
  DbView<PIX> v;
#pragma unroll
  for (int c = 0; c < 3; c++) { v.pl[c] = (PIX*)p.plane[c]; v.st[c] = p.stride[c]; v.ox[c] = 0; v.oy[c] = 0; }
  return v;
}

/* one edge unit (x4, y4) of the 8x8 luma grid: vertical edges at even x4, horizontal at even y4.  PART: 0 = its luma segment and (where it is on the chroma
   grid) its chroma segments, 1 = luma only, 2 = chroma only (k_deblock_tiles: luma and chroma tiles are cut at different places) */
template <class PIX, bool VERTICAL, int PART = 0>
__device__ __forceinline__ void d_deblock_unit(const DevPic& p, const DbView<PIX>& V, const int x4, const int y4)
{
  const int xDi = x4 << 2, yDi = y4 << 2;
  if ((VERTICAL && x4 == 0) || (!VERTICAL && y4 == 0)) return; /* picture border: never flagged */
  const int xp = VERTICAL ? xDi - 1 : xDi, yp = VERTICAL ? yDi : yDi - 1;
  /* tile sharding: an edge is computed by the owner(s) of its two sides, each writes only its own side */
  bool ownP = true, ownQ = true;
  if (p.ctb_owner) {
    ownQ = p.ctb_owner[d_ctb_of(p, xDi, yDi)] != 0; ownP = p.ctb_owner[d_ctb_of(p, xp, yp)] != 0;
    if (!ownP && !ownQ) return;
  }

  /* ---- memory round trip 1: the edge flags, the CU / PB indices of both sides and the CTB's slice index.  Segments that are no
     transform / prediction edge leave here (half of the 8x8 grid of a picture of mixed CU sizes). ---- */
  const int u = y4 * p.w4 + x4, uo = VERTICAL ? u - 1 : u - p.w4;
  const uint32_t ciQ = d_cu_index_at(p, xDi, yDi), ciP = d_cu_index_at(p, xp, yp);
  const int ef = p.edge_tu[u] | p.edge_pb[u], efo = p.edge_tu[uo];
  const uint32_t ip = p.pb_of[uo], iq = p.pb_of[u];
  const int slice_idx = p.ctbs[d_ctb_of(p, xDi, yDi)].slice_idx;
  /* all of the above is requested HERE: without the fence hipcc sinks every load but the edge flags below the exit — three dependent
     round trips (flags, indices, records) instead of two */
  M355_COMPILER_FENCE();
  if (!(ef & (VERTICAL ? (E_TU_V | E_PB_V) : (E_TU_H | E_PB_H)))) return;
  /* ---- round trip 2: the records the indices name AND the segment's luma samples, requested together (the pass used to be a
     chain of five dependent round trips with 32 scalar sample loads at its end): 32 + 32 bytes as aligned 4-sample vectors — the
     segment's 4 x 8 samples belong to this thread alone in this pass ---- */
  const int stride = V.st[0];
  PIX* const ptr = V.pl[0] + (yDi - V.oy[0]) * stride + (xDi - V.ox[0]);
  /* VERTICAL: rp[k] / rq[k] = line k (samples xDi-4 .. xDi-1 / xDi .. xDi+3); horizontal: rp[i] / rq[i] = the row at distance i
     from the edge (its four samples are the four lines) */
  Raw4<PIX> rp[4], rq[4];
  if (PART != 2) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
      rp[j] = d_ld4<PIX>(VERTICAL ? ptr + j * stride - 4 : ptr - (j + 1) * stride);
      rq[j] = d_ld4<PIX>(VERTICAL ? ptr + j * stride : ptr + j * stride);
    }
  }
#define PV(k, i) (VERTICAL ? d_get<PIX>(rp[k], 3 - (i)) : d_get<PIX>(rp[i], k))     /* line k, distance i */
#define QV(k, i) (VERTICAL ? d_get<PIX>(rq[k], i) : d_get<PIX>(rq[i], k))
#define SETP(k, i, v) do { if (VERTICAL) d_set<PIX>(rp[k], 3 - (i), v); else d_set<PIX>(rp[i], k, v); } while (0)
#define SETQ(k, i, v) do { if (VERTICAL) d_set<PIX>(rq[k], i, v); else d_set<PIX>(rq[i], k, v); } while (0)
  /* (an absent record reads the CTB table instead: always there, never used) */
  const bool pb_ok = ip && iq && ip <= (uint32_t)p.n_pb_records && iq <= (uint32_t)p.n_pb_records;
  const m355_cu cuQ = *(ciQ ? p.cus + (ciQ - 1) : (const m355_cu*)p.ctbs), cuP = *(ciP ? p.cus + (ciP - 1) : (const m355_cu*)p.ctbs);
  const m355_pb A = *(pb_ok ? p.pbs + (ip - 1) : (const m355_pb*)p.ctbs), B = *(pb_ok ? p.pbs + (iq - 1) : (const m355_pb*)p.ctbs);
  const m355_slice sh = p.slices[slice_idx];
  CuInfo Q = {0, 0, 0, 0}, P = {0, 0, 0, 0};
  if (ciQ) { Q.pred_mode = cuQ.pred_mode; Q.qp = cuQ.qp_y; Q.pcm = (cuQ.flags & M355_CUF_PCM) != 0; Q.bypass = (cuQ.flags & M355_CUF_TRANSQUANT_BYPASS) != 0; }
  if (ciP) { P.pred_mode = cuP.pred_mode; P.qp = cuP.qp_y; P.pcm = (cuP.flags & M355_CUF_PCM) != 0; P.bypass = (cuP.flags & M355_CUF_TRANSQUANT_BYPASS) != 0; }
  const int bS = d_boundary_strength(ef, efo, VERTICAL, P, Q, pb_ok, A, B);
  if (bS == 0) return;

  const bool plf = (p.pp.flags & M355_PF_PCM_LOOP_FILTER_DISABLE) != 0;
  const bool filterP = ownP && !((plf && P.pcm) || P.bypass), filterQ = ownQ && !((plf && Q.pcm) || Q.bypass);
  const int qP_L = (Q.qp + P.qp + 1) >> 1;

  /* ---- luma (deblock.cc:480-601, fallback-deblk.h:33-100) ---- */
  if (PART != 2) {
    const int bd = p.pp.bit_depth_luma;
    const int beta = c_tab_beta[d_clip3(0, 51, qP_L + sh.beta_offset)] * (1 << (bd - 8));
    const int tc = c_tab_tc[d_clip3(0, 53, qP_L + 2 * (bS - 1) + sh.tc_offset)] * (1 << (bd - 8));
    const int dp0 = d_abs(PV(0, 2) - 2 * PV(0, 1) + PV(0, 0)), dp3 = d_abs(PV(3, 2) - 2 * PV(3, 1) + PV(3, 0));
    const int dq0 = d_abs(QV(0, 2) - 2 * QV(0, 1) + QV(0, 0)), dq3 = d_abs(QV(3, 2) - 2 * QV(3, 1) + QV(3, 0));
    const int dpq0 = dp0 + dq0, dpq3 = dp3 + dq3, dp = dp0 + dp3, dq = dq0 + dq3, d = dpq0 + dpq3;
    if (d < beta) {
      const bool dSam0 = 2 * dpq0 < (beta >> 2) && d_abs(PV(0, 3) - PV(0, 0)) + d_abs(QV(0, 0) - QV(0, 3)) < (beta >> 3) &&
                         d_abs(PV(0, 0) - QV(0, 0)) < ((5 * tc + 1) >> 1);
      const bool dSam3 = 2 * dpq3 < (beta >> 2) && d_abs(PV(3, 3) - PV(3, 0)) + d_abs(QV(3, 0) - QV(3, 3)) < (beta >> 3) &&
                         d_abs(PV(3, 0) - QV(3, 0)) < ((5 * tc + 1) >> 1);
      const bool strong = dSam0 && dSam3;
      const bool dEp = dp < ((beta + (beta >> 1)) >> 3), dEq = dq < ((beta + (beta >> 1)) >> 3);
      /* the filtered samples replace the loaded ones; np / nq = how far from the edge a side was modified (0: not at all) */
      int np = 0, nq = 0;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int p0 = PV(k, 0), p1 = PV(k, 1), p2 = PV(k, 2), p3 = PV(k, 3);
        const int q0 = QV(k, 0), q1 = QV(k, 1), q2 = QV(k, 2), q3 = QV(k, 3);
        if (strong) {
          if (filterP) {
            SETP(k, 0, d_clip3(p0 - 2 * tc, p0 + 2 * tc, (p2 + 2 * p1 + 2 * p0 + 2 * q0 + q1 + 4) >> 3));
            SETP(k, 1, d_clip3(p1 - 2 * tc, p1 + 2 * tc, (p2 + p1 + p0 + q0 + 2) >> 2));
            SETP(k, 2, d_clip3(p2 - 2 * tc, p2 + 2 * tc, (2 * p3 + 3 * p2 + p1 + p0 + q0 + 4) >> 3));
            np = 3;
          }
          if (filterQ) {
            SETQ(k, 0, d_clip3(q0 - 2 * tc, q0 + 2 * tc, (p1 + 2 * p0 + 2 * q0 + 2 * q1 + q2 + 4) >> 3));
            SETQ(k, 1, d_clip3(q1 - 2 * tc, q1 + 2 * tc, (p0 + q0 + q1 + q2 + 2) >> 2));
            SETQ(k, 2, d_clip3(q2 - 2 * tc, q2 + 2 * tc, (p0 + q0 + q1 + 3 * q2 + 2 * q3 + 4) >> 3));
            nq = 3;
          }
        } else {
          int delta = (9 * (q0 - p0) - 3 * (q1 - p1) + 8) >> 4;
          if (d_abs(delta) < tc * 10) {
            delta = d_clip3(-tc, tc, delta);
            if (filterP) { SETP(k, 0, d_clip_bd(p0 + delta, bd)); np = max(np, 1); }
            if (filterQ) { SETQ(k, 0, d_clip_bd(q0 - delta, bd)); nq = max(nq, 1); }
            if (dEp && filterP) { SETP(k, 1, d_clip_bd(p1 + d_clip3(-(tc >> 1), tc >> 1, (((p2 + p0 + 1) >> 1) - p1 + delta) >> 1), bd)); np = max(np, 2); }
            if (dEq && filterQ) { SETQ(k, 1, d_clip_bd(q1 + d_clip3(-(tc >> 1), tc >> 1, (((q2 + q0 + 1) >> 1) - q1 - delta) >> 1), bd)); nq = max(nq, 2); }
          }
        }
      }
      /* write back whole vectors of the sides / rows that changed (nobody else touches this segment's samples in this pass) */
#pragma unroll
      for (int j = 0; j < 4; j++) {
        if (VERTICAL ? np > 0 : j < np) d_st4<PIX>(VERTICAL ? ptr + j * stride - 4 : ptr - (j + 1) * stride, rp[j]);
        if (VERTICAL ? nq > 0 : j < nq) d_st4<PIX>(ptr + j * stride, rq[j]);
      }
    }
  }
#undef PV
#undef QV
#undef SETP
#undef SETQ

  /* ---- chroma (deblock.cc:635-761): bS == 2 only, on the 8-sample chroma grid ---- */
  if (PART != 1 && bS > 1 && p.pp.chroma_format_idc != 0) {
    const int SW = p.sw, SH = p.sh;
    /* the reference visits x4 = 0, 2*SW, ... (vertical) resp. y4 = 0, 2*SH, ... and, along the edge,
       every SH-th (resp. SW-th) 4-luma unit */
    const bool on_grid = VERTICAL ? ((x4 % (2 * SW)) == 0 && (y4 % SH) == 0) : ((y4 % (2 * SH)) == 0 && (x4 % SW) == 0);
    if (on_grid) {
      const int bd = p.pp.bit_depth_chroma;
      const int xc = xDi / SW, yc = yDi / SH;
      const int stride = V.st[1];
      const int across = VERTICAL ? 1 : stride, along = VERTICAL ? stride : 1;
#pragma unroll
      for (int cp = 0; cp < 2; cp++) {
        const int qP_i = qP_L + (cp == 0 ? p.pp.pic_cb_qp_offset : p.pp.pic_cr_qp_offset);
        int QP_C;
        if (p.pp.chroma_format_idc == 1) QP_C = qP_i < 30 ? qP_i : (qP_i >= 43 ? qP_i - 6 : c_qpc_420[qP_i - 30]);
        else QP_C = min(qP_i, 51);
        const int tc = c_tab_tc[d_clip3(0, 53, QP_C + 2 * (bS - 1) + sh.tc_offset)] * (1 << (bd - 8));
        PIX* ptr = V.pl[cp + 1] + (yc - V.oy[cp + 1]) * stride + (xc - V.ox[cp + 1]);
#pragma unroll
        for (int k = 0; k < 4; k++) {
          PIX* o = ptr + k * along;
          const int p0 = o[-across], p1 = o[-2 * across], q0 = o[0], q1 = o[across];
          const int delta = d_clip3(-tc, tc, ((((q0 - p0) * 4) + p1 - q1 + 4) >> 3));
          if (filterP) o[-across] = (PIX)d_clip_bd(p0 + delta, bd);
          if (filterQ) o[0] = (PIX)d_clip_bd(q0 - delta, bd);
        }
      }
    }
  }
}

template <class PIX, bool VERTICAL>
__device__ __forceinline__ void k_deblock_body(const DevPic& p)
{
  M355_GATE(p);
  const int nx = VERTICAL ? (p.w4 + 1) / 2 : p.w4;
  const int ny = VERTICAL ? p.h4 : (p.h4 + 1) / 2;
  const int tx = blockIdx.x * 64 + (threadIdx.x & 63), ty = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (tx >= nx || ty >= ny) return;
  d_deblock_unit<PIX, VERTICAL>(p, d_db_view_picture<PIX>(p), VERTICAL ? tx * 2 : tx, VERTICAL ? ty : ty * 2);
}
template <class PIX, bool VERTICAL> __global__ void __launch_bounds__(256) k_deblock(DevPic p) { k_deblock_body<PIX, VERTICAL>(p); }
template <class PIX, bool VERTICAL> __global__ void __launch_bounds__(256) k_deblock_batch(DevBatch b) { M355_BATCH_PIC(b); k_deblock_body<PIX, VERTICAL>(p); }

/* ---- both directions in ONE pass (unsharded 4:2:0 / monochrome pictures) ----
 * A workgroup owns the 64x64 luma samples [64 i - 4, 64 i + 60) x [64 j - 4, 64 j + 60), i.e. a tile shifted by (-4, -4) against the 8x8 edge grid: every
 * luma edge of either direction whose first sample lies in the tile has its WHOLE reach in it (4 read / 3 modified samples on each side of edges that are
 * multiples of 8 apart), and every sample of the tile is modified only by edges of the tile — no halo, nothing filtered twice.  Chroma (4:2:0) is cut the
 * same way in ITS OWN units: the 32x32 chroma samples [32 i - 4, 32 i + 28) x [32 j - 4, 32 j + 28) — chroma edges lie on an 8-sample chroma grid, a chroma
 * segment is four chroma lines (= two luma units) long, so the cut has to fall on a multiple of 4 chroma samples that is not a multiple of 8.  The tiles are
 * fetched into LDS once, all their vertical edges are filtered there (d_deblock_unit on an LDS view: the reference's "all vertical edges first",
 * deblock.cc:919-939, per tile), a barrier, all horizontal edges, a barrier, and only the 4-sample (chroma: 2-sample) vectors that changed are written back:
 * the picture is read once instead of twice and there is one launch instead of two. */
template <class PIX>
__global__ void __launch_bounds__(128) k_deblock_tiles(DevPic p)
{
  M355_GATE(p);
  const bool chroma = p.pp.chroma_format_idc == 1;
  const int bi = (int)blockIdx.x, bj = (int)blockIdx.y;
  /* luma tile, chroma tile (chroma samples); either may be empty in the last row / column of workgroups */
  const int x0 = max(bi * 64 - 4, 0), y0 = max(bj * 64 - 4, 0), x1 = min(bi * 64 + 60, p.pw[0]), y1 = min(bj * 64 + 60, p.ph[0]);
  const int cx0 = max(bi * 32 - 4, 0), cy0 = max(bj * 32 - 4, 0), cx1 = chroma ? min(bi * 32 + 28, p.pw[1]) : 0, cy1 = chroma ? min(bj * 32 + 28, p.ph[1]) : 0;
  const bool has_l = x0 < x1 && y0 < y1, has_c = cx0 < cx1 && cy0 < cy1;
  if (!has_l && !has_c) return;
  __shared__ __attribute__((aligned(16))) PIX s_l[64 * 64];
  __shared__ __attribute__((aligned(16))) PIX s_c[2][32 * 32];
  DbView<PIX> V;
  V.pl[0] = s_l; V.st[0] = 64; V.ox[0] = x0; V.oy[0] = y0;
  V.pl[1] = s_c[0]; V.pl[2] = s_c[1]; V.st[1] = V.st[2] = 32; V.ox[1] = V.ox[2] = cx0; V.oy[1] = V.oy[2] = cy0;
  const int t = (int)threadIdx.x;
  /* fetch: luma as 4-sample vectors (16 per row), chroma as 2-sample pairs (16 per row and plane); the originals stay in registers for the write-back */
  const int wv = has_l ? (x1 - x0) >> 2 : 0, hl = has_l ? y1 - y0 : 0, wp = has_c ? (cx1 - cx0) >> 1 : 0, hc = has_c ? cy1 - cy0 : 0;
  Raw4<PIX> ol[8];
  uint32_t oc[2][4];
  {
    const PIX* gl = (const PIX*)p.plane[0] + (size_t)y0 * p.stride[0] + x0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const int idx = t + 128 * k, r = idx >> 4, v = idx & 15;
      ol[k].w[0] = 0; ol[k].w[sizeof(PIX) == 2 ? 1 : 0] = 0;
      if (r < hl && v < wv) { ol[k] = d_ld4<PIX>(gl + (size_t)r * p.stride[0] + 4 * v); d_st4<PIX>(s_l + r * 64 + 4 * v, ol[k]); }
    }
  }
#pragma unroll
  for (int c = 0; c < 2; c++) {
    const PIX* gc = (const PIX*)p.plane[c + 1] + (size_t)cy0 * p.stride[c + 1] + cx0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int idx = t + 128 * k, r = idx >> 4, v = idx & 15;
      oc[c][k] = 0;
      if (r < hc && v < wp) {
        const PIX* g = gc + (size_t)r * p.stride[c + 1] + 2 * v;
        oc[c][k] = sizeof(PIX) == 2 ? (uint32_t)d_ldg4((const M355_GLOBAL void*)g) : (uint32_t)d_ldg2((const M355_GLOBAL void*)g);
        if (sizeof(PIX) == 2) *(uint32_t*)(s_c[c] + r * 32 + 2 * v) = oc[c][k]; else *(uint16_t*)(s_c[c] + r * 32 + 2 * v) = (uint16_t)oc[c][k];
      }
    }
  }
  __syncthreads();
  /* vertical edges: the luma tile's 8 edges x 16 units, one per thread; the chroma tile's 4 edges x 8 segments (luma units x4 = 16 i + 4 a, y4 = 16 j - 2 + 2 b) */
  {
    const int x4 = (((x0 >> 2) + 1) & ~1) + 2 * (t & 7), y4 = (y0 >> 2) + (t >> 3);
    if (has_l && 4 * x4 + 4 <= x1 && 4 * y4 < y1) d_deblock_unit<PIX, true, 1>(p, V, x4, y4);
    const int cx4 = 16 * bi + 4 * (t & 3), cy4 = 16 * bj - 2 + 2 * ((t >> 2) & 7);
    if (has_c && t < 32 && cy4 >= 0 && 2 * cx4 - 2 >= cx0 && 2 * cx4 + 2 <= cx1 && 2 * cy4 >= cy0 && 2 * cy4 + 4 <= cy1) d_deblock_unit<PIX, true, 2>(p, V, cx4, cy4);
  }
  __syncthreads();
  {
    const int x4 = (x0 >> 2) + (t & 15), y4 = (((y0 >> 2) + 1) & ~1) + 2 * (t >> 4);
    if (has_l && 4 * x4 < x1 && 4 * y4 + 4 <= y1) d_deblock_unit<PIX, false, 1>(p, V, x4, y4);
    const int cy4 = 16 * bj + 4 * (t & 3), cx4 = 16 * bi - 2 + 2 * ((t >> 2) & 7);
    if (has_c && t < 32 && cx4 >= 0 && 2 * cy4 - 2 >= cy0 && 2 * cy4 + 2 <= cy1 && 2 * cx4 >= cx0 && 2 * cx4 + 4 <= cx1) d_deblock_unit<PIX, false, 2>(p, V, cx4, cy4);
  }
  __syncthreads();
  /* write-back: the vectors that changed */
  {
    PIX* wl = (PIX*)p.plane[0] + (size_t)y0 * p.stride[0] + x0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const int idx = t + 128 * k, r = idx >> 4, v = idx & 15;
      if (r < hl && v < wv) {
        const Raw4<PIX> n = d_ld4<PIX>(s_l + r * 64 + 4 * v);
        if (n.w[0] != ol[k].w[0] || n.w[sizeof(PIX) == 2 ? 1 : 0] != ol[k].w[sizeof(PIX) == 2 ? 1 : 0]) d_st4<PIX>(wl + (size_t)r * p.stride[0] + 4 * v, n);
      }
    }
  }
#pragma unroll
  for (int c = 0; c < 2; c++) {
    PIX* wc = (PIX*)p.plane[c + 1] + (size_t)cy0 * p.stride[c + 1] + cx0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int idx = t + 128 * k, r = idx >> 4, v = idx & 15;
      if (r < hc && v < wp) {
        const uint32_t n = sizeof(PIX) == 2 ? *(const uint32_t*)(s_c[c] + r * 32 + 2 * v) : (uint32_t)*(const uint16_t*)(s_c[c] + r * 32 + 2 * v);
        if (n != oc[c][k]) {
          PIX* g = wc + (size_t)r * p.stride[c + 1] + 2 * v;
          if (sizeof(PIX) == 2) d_stg4((M355_GLOBAL void*)g, n); else *(uint16_t*)g = (uint16_t)n;
        }
      }
    }
  }
}

template <class PIX>
static void launch_pass_batch(const HostBatch& b, bool vertical, hipStream_t st)
{
  int w4 = 0, h4 = 0;
  for (int k = 0; k < b.n; k++) if ((b.on >> k) & 1u) { w4 = std::max(w4, b.host[k].w4); h4 = std::max(h4, b.host[k].h4); }
  if (!w4) return;
  const DevBatch d{b.dev, b.on};
  if (vertical) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_deblock_batch<PIX, true>), dim3(((w4 + 1) / 2 + 63) / 64, (h4 + 3) / 4, b.n), dim3(256), 0, st, d);
  else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_deblock_batch<PIX, false>), dim3((w4 + 63) / 64, ((h4 + 1) / 2 + 3) / 4, b.n), dim3(256), 0, st, d);
}
void m355_launch_deblock_batch(const HostBatch& b, bool hbd, hipStream_t st)
{
  for (int v = 1; v >= 0; v--) { if (hbd) launch_pass_batch<uint16_t>(b, v != 0, st); else launch_pass_batch<uint8_t>(b, v != 0, st); }
}

template <class PIX>
static void launch_pass(const DevPic& p, bool vertical, hipStream_t st)
{
  if (vertical) {
    const int nx = (p.w4 + 1) / 2, ny = p.h4;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_deblock<PIX, true>), dim3((nx + 63) / 64, (ny + 3) / 4), dim3(256), 0, st, p);
  } else {
    const int nx = p.w4, ny = (p.h4 + 1) / 2;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_deblock<PIX, false>), dim3((nx + 63) / 64, (ny + 3) / 4), dim3(256), 0, st, p);
  }
}

void m355_launch_deblock_pass(const DevPic& p, bool hbd, bool vertical, hipStream_t st)
{
  if (hbd) launch_pass<uint16_t>(p, vertical, st);
  else launch_pass<uint8_t>(p, vertical, st);
}

void m355_launch_deblock(const DevPic& p, bool hbd, hipStream_t st)
{
#ifndef M355_DEBLOCK_TWO_PASS
  if (p.pp.chroma_format_idc <= 1 && !p.ctb_owner) {            /* both directions in one pass over shifted tiles */
    const dim3 grid((p.pw[0] + 8 + 63) / 64, (p.ph[0] + 8 + 63) / 64);     /* (the chroma tiles reach 8 luma samples further than the luma tiles) */
    if (hbd) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_deblock_tiles<uint16_t>), grid, dim3(128), 0, st, p);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_deblock_tiles<uint8_t>), grid, dim3(128), 0, st, p);
    return;
  }
#endif
  m355_launch_deblock_pass(p, hbd, true, st);
  m355_launch_deblock_pass(p, hbd, false, st);
}
