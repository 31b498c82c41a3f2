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

template <class PIX, bool VERTICAL>
__device__ __forceinline__ void k_deblock_body(const DevPic& p)
{
  M355_GATE(p);
  /* thread -> edge unit on the 8x8 luma grid: vertical edges at even x4, horizontal at even y4 */
  const int nx = VERTICAL ? (p.w4 + 1) / 2 : p.w4;
  const int ny = VERTICAL ? p.h4 : (p.h4 + 1) / 2;
  const int tx = blockIdx.x * 64 + (threadIdx.x & 63), ty = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (tx >= nx || ty >= ny) return;
  const int x4 = VERTICAL ? tx * 2 : tx, y4 = VERTICAL ? ty : ty * 2;
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
  const int stride = p.stride[0];
  PIX* const ptr = (PIX*)p.plane[0] + yDi * stride + xDi;
  /* VERTICAL: rp[k] / rq[k] = line k (samples xDi-4 .. xDi-1 / xDi .. xDi+3); horizontal: rp[i] / rq[i] = the row at distance i
     from the edge (its four samples are the four lines) */
  Raw4<PIX> rp[4], rq[4];
#pragma unroll
  for (int j = 0; j < 4; j++) {
    rp[j] = d_ld4<PIX>(VERTICAL ? ptr + j * stride - 4 : ptr - (j + 1) * stride);
    rq[j] = d_ld4<PIX>(VERTICAL ? ptr + j * stride : ptr + j * stride);
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
  {
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
  if (bS > 1 && p.pp.chroma_format_idc != 0) {
    const int SW = p.sw, SH = p.sh;
    /* the reference visits x4 = 0, 2*SW, ... (vertical) resp. y4 = 0, 2*SH, ... and, along the edge,
       every SH-th (resp. SW-th) 4-luma unit */
    const bool on_grid = VERTICAL ? ((x4 % (2 * SW)) == 0 && (y4 % SH) == 0) : ((y4 % (2 * SH)) == 0 && (x4 % SW) == 0);
    if (on_grid) {
      const int bd = p.pp.bit_depth_chroma;
      const int xc = xDi / SW, yc = yDi / SH;
      const int stride = p.stride[1];
      const int across = VERTICAL ? 1 : stride, along = VERTICAL ? stride : 1;
#pragma unroll
      for (int cp = 0; cp < 2; cp++) {
        const int qP_i = qP_L + (cp == 0 ? p.pp.pic_cb_qp_offset : p.pp.pic_cr_qp_offset);
        int QP_C;
        if (p.pp.chroma_format_idc == 1) QP_C = qP_i < 30 ? qP_i : (qP_i >= 43 ? qP_i - 6 : c_qpc_420[qP_i - 30]);
        else QP_C = min(qP_i, 51);
        const int tc = c_tab_tc[d_clip3(0, 53, QP_C + 2 * (bS - 1) + sh.tc_offset)] * (1 << (bd - 8));
        PIX* ptr = (PIX*)p.plane[cp + 1] + yc * stride + xc;
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

template <class PIX, bool VERTICAL> __global__ void __launch_bounds__(256) k_deblock(DevPic p) { k_deblock_body<PIX, VERTICAL>(p); }
template <class PIX, bool VERTICAL> __global__ void __launch_bounds__(256) k_deblock_batch(DevBatch b) { M355_BATCH_PIC(b); k_deblock_body<PIX, VERTICAL>(p); }

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
  m355_launch_deblock_pass(p, hbd, true, st);
  m355_launch_deblock_pass(p, hbd, false, st);
}
