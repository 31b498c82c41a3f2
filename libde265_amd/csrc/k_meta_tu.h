/*
 * k_meta_tu.h — the transform-edge scatter of k_meta.hip as a body function of its own: k_meta.hip launches it alone (k_meta_tu), k_intra.hip
 * beside the border plans in one launch (k_tu_plan, M355_MERGE_TU_PLAN: both only read what k_meta_planes wrote).
 */
#ifndef M355_K_META_TU_H
#define M355_K_META_TU_H
#include "k_common.h"

/* one thread per transform-tree leaf: transform edges + cbf_luma (deblock.cc:33-63, slice.cc:2958) */
__device__ __forceinline__ void k_meta_tu_body(const DevPic& p, const int blk)
{
  M355_GATE(p);
  const int i = blk * blockDim.x + threadIdx.x;
  if (i >= p.n_tus) return;
  const m355_tu tu = p.tus[i];
  const int n4 = (1 << tu.log2_size) >> 2;
  const int ux0 = tu.x >> 2, uy0 = tu.y >> 2;
  const uint32_t ci = d_cu_index_at(p, tu.x, tu.y);
  int left = 0, top = 0;
  if (ci) {
    const m355_cu cu = p.cus[ci - 1];
    const uint8_t f = p.cuf[ci - 1];
    if (f & 4) {
      left = (tu.x == cu.x) ? (f & 1) : 1;
      top = (tu.y == cu.y) ? ((f >> 1) & 1) : 1;
    }
  }
  const int nz = (tu.flags & M355_TUF_NONZERO_COEFF) ? E_NONZERO : 0;
  if (nz) {
    for (int y = 0; y < n4 && uy0 + y < p.h4; y++)
      for (int x = 0; x < n4 && ux0 + x < p.w4; x++)
        p.edge_tu[(uy0 + y) * p.w4 + ux0 + x] = (uint8_t)(nz | ((x == 0 && left) ? E_TU_V : 0) | ((y == 0 && top) ? E_TU_H : 0));
  } else {
    if (left)
      for (int y = 0; y < n4 && uy0 + y < p.h4; y++)
        p.edge_tu[(uy0 + y) * p.w4 + ux0] = (uint8_t)(E_TU_V | ((y == 0 && top) ? E_TU_H : 0));
    if (top)
      for (int x = (left ? 1 : 0); x < n4 && ux0 + x < p.w4; x++) p.edge_tu[uy0 * p.w4 + ux0 + x] = E_TU_H;
  }
}

#endif
