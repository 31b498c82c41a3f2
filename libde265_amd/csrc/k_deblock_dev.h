/*
 * k_deblock_dev.h — device helpers of the deblocking filter shared by k_deblock.hip (the two passes) and k_sao.hip (the
 * EXPERIMENTAL horizontal-edge pass inside the SAO kernel, M355_FUSE_DBH): the beta / tc / chroma-QP tables (deblock.cc:385-410,
 * transform.h:29-34), derive_boundaryStrength on fetched values (deblock.cc:243-383) and the raw 4-sample row vectors.
 */
#ifndef K_DEBLOCK_DEV_H
#define K_DEBLOCK_DEV_H
#include "k_common.h"

/* k_deblock.hip (K_DEBLOCK_DEV_OWNER) defines the tables with external linkage, as it always has; any other translation unit gets a
   copy of its own (the library is linked without relocatable device code) */
#ifndef K_DEBLOCK_DEV_OWNER
namespace {
#endif
__constant__ uint8_t c_tab_beta[52] = {0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  6,  7,
                                       8,  9,  10, 11, 12, 13, 14, 15, 16, 17, 18, 20, 22, 24, 26, 28, 30, 32,
                                       34, 36, 38, 40, 42, 44, 46, 48, 50, 52, 54, 56, 58, 60, 62, 64};
__constant__ uint8_t c_tab_tc[54] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,  0,  0,  0,  0,
                                     1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 3,  3,  3,  3,  4,
                                     4, 4, 5, 5, 6, 6, 7, 8, 9, 10, 11, 13, 14, 16, 18, 20, 22, 24};
__constant__ int8_t c_qpc_420[14] = {29, 30, 31, 32, 33, 33, 34, 34, 35, 35, 36, 36, 37, 37}; /* transform.h:29-34 */

#ifndef K_DEBLOCK_DEV_OWNER
}  /* namespace */
#endif

struct CuInfo { int pred_mode, qp, pcm, bypass; };
__device__ __forceinline__ CuInfo d_cu_info(const DevPic& p, int xl, int yl)
{
  CuInfo r = {0, 0, 0, 0};
  const uint32_t ci = d_cu_index_at(p, xl, yl);
  if (ci) {
    const m355_cu cu = p.cus[ci - 1];
    r.pred_mode = cu.pred_mode; r.qp = cu.qp_y;
    r.pcm = (cu.flags & M355_CUF_PCM) != 0; r.bypass = (cu.flags & M355_CUF_TRANSQUANT_BYPASS) != 0;
  }
  return r;
}

/* derive_boundaryStrength for one edge unit (deblock.cc:243-383) on values the caller has fetched: ef = the unit's edge flags
   (TU | PB), efo = the TU flags of the unit across the edge, A / B = the prediction blocks across / on this side (pb_ok: both
   exist in this picture's records) */
__device__ __forceinline__ int d_boundary_strength(int ef, int efo, bool vertical, const CuInfo& P, const CuInfo& Q, bool pb_ok, const m355_pb& A, const m355_pb& B)
{
  const int edgeMask = vertical ? (E_TU_V | E_PB_V) : (E_TU_H | E_PB_H);
  if (!(ef & edgeMask)) return 0;
  if (P.pred_mode == 0 || Q.pred_mode == 0) return 2;
  if ((ef & (vertical ? E_TU_V : E_TU_H)) && ((ef & E_NONZERO) || (efo & E_NONZERO))) return 1;
  /* pb_of is not cleared between pictures (k_meta.hip): an inter CU whose units no PB of THIS picture covers (a list that
     validation cannot fully check) leaves a stale index — never followed past this picture's records (pb_ok) */
  if (!pb_ok) return 0;
  const bool pf0 = A.flags & M355_PBF_PRED_L0, pf1 = A.flags & M355_PBF_PRED_L1;
  const bool qf0 = B.flags & M355_PBF_PRED_L0, qf1 = B.flags & M355_PBF_PRED_L1;
  const int rP0 = pf0 ? A.ref_slot[0] : -1, rP1 = pf1 ? A.ref_slot[1] : -1;
  const int rQ0 = qf0 ? B.ref_slot[0] : -1, rQ1 = qf1 ? B.ref_slot[1] : -1;
  if (!((rP0 == rQ0 && rP1 == rQ1) || (rP0 == rQ1 && rP1 == rQ0))) return 1;
  const int p0x = pf0 ? A.mv[0][0] : 0, p0y = pf0 ? A.mv[0][1] : 0, p1x = pf1 ? A.mv[1][0] : 0, p1y = pf1 ? A.mv[1][1] : 0;
  const int q0x = qf0 ? B.mv[0][0] : 0, q0y = qf0 ? B.mv[0][1] : 0, q1x = qf1 ? B.mv[1][0] : 0, q1y = qf1 ? B.mv[1][1] : 0;
#define FAR(ax, ay, bx, by) (d_abs((ax) - (bx)) >= 4 || d_abs((ay) - (by)) >= 4)
  if (rP0 != rP1) {
    if (rP0 == rQ0) return (FAR(p0x, p0y, q0x, q0y) || FAR(p1x, p1y, q1x, q1y)) ? 1 : 0;
    return (FAR(p0x, p0y, q1x, q1y) || FAR(p1x, p1y, q0x, q0y)) ? 1 : 0;
  }
  return ((FAR(p0x, p0y, q0x, q0y) || FAR(p1x, p1y, q1x, q1y)) && (FAR(p0x, p0y, q1x, q1y) || FAR(p1x, p1y, q0x, q0y))) ? 1 : 0;
#undef FAR
}

/* Four adjacent samples = one aligned vector (8 bytes of uint16, 4 bytes of uint8), kept RAW in registers: the segment's
   4 x 8 samples are 16 (8) dwords instead of 32 unpacked ones, a sample is a bit-field extract / insert at a compile-time
   position (all loops below are unrolled) — 8 waves per SIMD for both directions. */
template <class PIX> struct Raw4 { uint32_t w[sizeof(PIX) == 2 ? 2 : 1]; };
template <class PIX> __device__ __forceinline__ Raw4<PIX> d_ld4(const PIX* q)
{
  Raw4<PIX> r;
  if (sizeof(PIX) == 2) { const uint2 v = *(const uint2*)q; r.w[0] = v.x; r.w[sizeof(PIX) == 2 ? 1 : 0] = v.y; }
  else r.w[0] = *(const uint32_t*)q;
  return r;
}
template <class PIX> __device__ __forceinline__ void d_st4(PIX* q, const Raw4<PIX>& r)
{
  if (sizeof(PIX) == 2) *(uint2*)q = make_uint2(r.w[0], r.w[sizeof(PIX) == 2 ? 1 : 0]);
  else *(uint32_t*)q = r.w[0];
}
template <class PIX> __device__ __forceinline__ int d_get(const Raw4<PIX>& r, int s)
{
  if (sizeof(PIX) == 2) return (int)((r.w[s >> 1] >> (16 * (s & 1))) & 0xFFFFu);
  return (int)((r.w[0] >> (8 * s)) & 0xFFu);
}
template <class PIX> __device__ __forceinline__ void d_set(Raw4<PIX>& r, int s, int v)
{
  if (sizeof(PIX) == 2) { const uint32_t m = 0xFFFFu << (16 * (s & 1)); r.w[s >> 1] = (r.w[s >> 1] & ~m) | ((uint32_t)v << (16 * (s & 1))); }
  else { const uint32_t m = 0xFFu << (8 * s); r.w[0] = (r.w[0] & ~m) | ((uint32_t)v << (8 * s)); }
}

#endif
