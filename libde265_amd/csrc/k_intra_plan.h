/*
 * k_intra_plan.h — the PLANNER of k_intra.hip as device functions of their own: the geometry of a CTB's LDS arrays (what a plan entry indexes) and the plan
 * of one block / one CTB.  Launched alone (k_intra_plan, k_intra_plan_batch), beside the transform-edge scatter (k_tu_plan), inside k_intra's prologue (one intra
 * picture at a time) and as a role of the residual launch of a one-stream lane (k_residual.hip: k_residual_tu_plan).
 */
#ifndef M355_K_INTRA_PLAN_H
#define M355_K_INTRA_PLAN_H
#include "k_common.h"

#define MAXCTB 64
#define M355_INTRA_KEEPER_NW 13   /* k_intra's instantiation for ONE intra picture at a time: the worker waves + the halo keeper; plans its CTBs itself */
/* body rows of a component: CTB width + 8 samples — sample x lives at column x + 8 (16-byte aligned 8-sample vectors).
   Chroma bodies are sized for the chroma format (template parameter CF): the LDS footprint decides how many CTBs a CU
   works on at once, and this stage lives on concurrency. */
#define BODY_PITCH_OF(cw) ((cw) + 8)
#define BODY_X0 8
#define SPIN_LIMIT M355_SPIN_LIMIT   /* k_asm.h: bound on the polls for one granule (a list that promises a sample nobody produces) */
/* A component's samples in LDS, ONE array of 16-bit elements: the body (above), then its halo — the row above the CTB, x = -1 ..
   2cw-1 at halo index x + 1, then the column left of it, y at halo index HALO_TOP_N + y; an element is a sample, or
   HALO_NOT_READY while the neighbour CTB has not published it — then one cell that holds 1 << (bitDepth - 1) (what a border is made
   of when nothing is available, intrapred.h:645-649).  A plan entry (k_intra_plan) is an element index into this array: where
   border entry e of a block comes from, substitution already applied. */
#define HALO_TOP_N (2 * MAXCTB + 2)
#define HALO_N (HALO_TOP_N + MAXCTB)
#define HALO_NOT_READY 0xFFFFu       /* (a 16-bit-deep picture may hold this value as a sample: the poll then succeeds at once) */
#define COMP_LDS(body) ((body) + HALO_N + 8)   /* elements per component: body + halo + constant cell (+ alignment) */

/* z-scan order inside a CTB (pps.cc:608-623 MinTbAddrZS, low bits): Morton code of the min-TB coordinates */
__device__ __forceinline__ uint32_t d_spread4(uint32_t v) { v = (v | (v << 2)) & 0x33u; return (v | (v << 1)) & 0x55u; }
__device__ __forceinline__ uint32_t d_morton(uint32_t x, uint32_t y) { return d_spread4(x) | (d_spread4(y) << 1); }

__device__ __forceinline__ bool d_is_intra_at(const DevPic& p, int xl, int yl)
{
  const uint32_t ci = d_cu_index_at(p, xl, yl);
  return ci == 0 || p.cus[ci - 1].pred_mode == 0; /* zero-initialised cb_info reads MODE_INTRA */
}

/* source entry for reference_sample_substitution (intrapred.h:637-665): nearest available entry
 * below e in scan order, else the lowest available entry */
__device__ __forceinline__ int d_subst_src(int e, unsigned long long m0, unsigned long long m1, unsigned long long m2)
{
  const int k = e >> 6, b = e & 63;
  const unsigned long long cur = k == 0 ? m0 : (k == 1 ? m1 : m2);
  const unsigned long long below = cur & ((1ull << b) - 1ull);
  if (below) return k * 64 + 63 - __clzll(below);
  if (k >= 2 && m1) return 64 + 63 - __clzll(m1);
  if (k >= 1 && m0) return 63 - __clzll(m0);
  if (m0) return __ffsll(m0) - 1;
  if (m1) return 64 + __ffsll(m1) - 1;
  return 128;
}

/* chroma CTB geometry of a chroma format */
template <int CF> struct IntraGeo {
  static constexpr int CW_C = (CF == 1 || CF == 2) ? MAXCTB / 2 : MAXCTB, CH_C = CF == 1 ? MAXCTB / 2 : MAXCTB;
  static constexpr int BODY_L = MAXCTB * BODY_PITCH_OF(MAXCTB), BODY_C = CF == 0 ? 8 : CH_C * BODY_PITCH_OF(CW_C);
  static constexpr int SAMP_L = COMP_LDS(BODY_L), SAMP_C = COMP_LDS(BODY_C);   /* elements of a luma / chroma component's array */
};

#ifndef PLAN_SPLIT
#define PLAN_SPLIT 8   /* (4 -> 8: C2 waits 15 us less for its plans, profiles/r03_u_*) */
#endif
/* the plan of ONE block (record k of the CTB's sorted list): `codes` = 4 * 32 + 8 elements of the calling wave's LDS scratch, `dst` = where
   the CTB's plans start (global memory: k_intra_plan; LDS: k_intra plans an intra picture's CTB itself, in its prologue) */
template <int CF>
__device__ __forceinline__ void d_intra_plan_block(const DevPic& p, const uint32_t ib_index, const int ctbX, const int ctbY, const uint32_t nb_same, const uint32_t nb_earlier, uint16_t* codes, uint16_t* dst)
{
  const int lane = threadIdx.x & 63;
  const int l2c = p.pp.log2_ctb_size;
  {
    const uint32_t* r = (const uint32_t*)&p.ibs[ib_index];
    const uint32_t w0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)r[0]), w1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)r[1]);
    const uint32_t aux = (uint32_t)__builtin_amdgcn_readfirstlane((int)p.ib_aux[4 * ib_index + 3]);
    const int c = (int)(w1 & 0xFFu), log2 = (int)((w1 >> 8) & 0xFFu), flags = (int)(w1 >> 24);
    if (flags & M355_IBF_PCM) return;                        /* raw blocks read no border: no plan entries */
    const int nT = 1 << log2;
    const int csw = c ? (p.sw == 2) : 0, csh = c ? (p.sh == 2) : 0;
    const int SubW = 1 << csw, SubH = 1 << csh;
    const int x0c = (ctbX << l2c) >> csw, y0c = (ctbY << l2c) >> csh;
    const int BODY_PITCH = c == 0 ? BODY_PITCH_OF(MAXCTB) : BODY_PITCH_OF(IntraGeo<CF>::CW_C);
    const int HALO_BASE = c == 0 ? IntraGeo<CF>::BODY_L : IntraGeo<CF>::BODY_C;   /* first halo element of the component's array */
    const int xB = (int)(w0 & 0xFFFFu), yB = (int)(w0 >> 16), lx = xB - x0c, ly = yB - y0c;
    /* ---- preproc (intrapred.h:436-531): CTB-level availability from the neighbourhood table ---- */
    const int xBL = xB * SubW, yBL = yB * SubH;
    bool aL = xBL != 0, aT = yBL != 0, aTL = xBL != 0 && yBL != 0, aTR = yBL != 0;
    if (xBL + nT * SubW >= p.pp.width) aTR = false;
    {
      const int dxL = ((xBL - 1) >> l2c) - ctbX, dxR = ((xBL + nT * SubW) >> l2c) - ctbX, dyT = ((yBL - 1) >> l2c) - ctbY;
      if (aL && !((nb_same >> (3 + dxL + 1)) & 1u)) aL = false;
      if (aT && !((nb_same >> ((dyT + 1) * 3 + 1)) & 1u)) aT = false;
      if (aTL && !((nb_same >> ((dyT + 1) * 3 + dxL + 1)) & 1u)) aTL = false;
      if (aTR && !((nb_same >> ((dyT + 1) * 3 + dxR + 1)) & 1u)) aTR = false;
    }
    int nBottom = p.pp.height - yB * SubH;
    nBottom = (nBottom + SubH - 1) >> csh;
    if (nBottom > 2 * nT) nBottom = 2 * nT;
    int nRight = p.pp.width - xB * SubW;
    nRight = (nRight + SubW - 1) >> csw;
    if (nRight > 2 * nT) nRight = 2 * nT;
    const int l2tb = p.pp.log2_min_tb_size, cmask = (1 << l2c) - 1;
    const uint32_t curZ = d_morton((uint32_t)(xBL & cmask) >> l2tb, (uint32_t)(yBL & cmask) >> l2tb);
    const bool cip = (p.pp.flags & M355_PF_CONSTRAINED_INTRA_PRED) != 0;
    const int nEnt = 4 * nT + 1;
    /* ---- fill_from_image (intrapred.h:534-633): availability and LDS source of every border entry ---- */
    unsigned long long am[3] = {0, 0, 0};
    uint32_t code[3] = {0, 0, 0};
#pragma unroll
    for (int q = 0; q < 3; q++) {
      if (64 * q >= nEnt) continue;              /* wave-uniform: 4x4 / 8x8 blocks have 17 / 33 border entries */
      const int e = lane + 64 * q;
      bool av = false;
      if (e < nEnt) {
        const int i = e - 2 * nT;
        int xN, yN, sx, sy; /* test position (luma), sample position (local) */
        if (i < 0) {
          const int yy = -i - 1, g4 = yy & ~3;
          av = aL && (g4 + 3 < nBottom);
          xN = (xB - 1) * SubW; yN = (yB + g4 + 3) * SubH; sx = lx - 1; sy = ly + yy;
        } else if (i == 0) {
          av = aTL;
          xN = (xB - 1) * SubW; yN = (yB - 1) * SubH; sx = lx - 1; sy = ly - 1;
        } else {
          const int xx = i - 1, g4 = xx & ~3;
          av = (g4 < nT ? aT : aTR) && (g4 < nRight);
          xN = (xB + g4) * SubW; yN = (yB - 1) * SubH; sx = lx + xx; sy = ly - 1;
        }
        if (av) {     /* MinTbAddrZS[neighbour] <= MinTbAddrZS[current] (intrapred.h:560-566) */
          const int dcx = (xN >> l2c) - ctbX, dcy = (yN >> l2c) - ctbY;
          if (dcx == 0 && dcy == 0) av = d_morton((uint32_t)(xN & cmask) >> l2tb, (uint32_t)(yN & cmask) >> l2tb) <= curZ;
          else av = (nb_earlier >> ((dcy + 1) * 3 + dcx + 1)) & 1u;
        }
        if (av && cip) av = d_is_intra_at(p, xN, yN);
        if (av) code[q] = sy < 0 ? (uint32_t)(HALO_BASE + sx + 1) : (sx < 0 ? (uint32_t)(HALO_BASE + HALO_TOP_N + sy) : (uint32_t)(sy * BODY_PITCH + sx + BODY_X0));
      }
      am[q] = __ballot(av);
    }
    /* ---- reference_sample_substitution (intrapred.h:637-665), on the sources ---- */
    const bool none = (am[0] | am[1] | am[2]) == 0;
#pragma unroll
    for (int q = 0; q < 3; q++) {
      if (64 * q >= nEnt) continue;
      const int e = lane + 64 * q;
      if (e < nEnt) codes[e] = (uint16_t)code[q];
    }
    wave_sync();
    uint16_t* out = dst + (aux & 0xFFFFu);
    /* entries the block's mode never reads (k_common.h m355_intra_used_entries) are pointed at the constant cell AFTER the
       substitution (an entry inside the used range may well take its value from one outside): k_intra's blocks fetch all 4nT + 1
       entries, and a HALO entry whose CTB has not published it yet would make the block wait for a sample it does not use */
    int top_e, left_e;
    m355_intra_used_entries((int)((w1 >> 16) & 0xFFu), log2, c, CF, p.pp.flags, (uint32_t)flags, &top_e, &left_e);
#pragma unroll
    for (int q = 0; q < 3; q++) {
      if (64 * q >= nEnt) continue;
      const int e = lane + 64 * q;
      if (e < nEnt) {
        const int i = e - 2 * nT;
        uint32_t v;
        if (none || i > top_e || i < -left_e) v = (uint32_t)(HALO_BASE + HALO_N);        /* the constant cell */
        else if ((am[q] >> lane) & 1) v = code[q];
        else v = codes[d_subst_src(e, am[0], am[1], am[2])];
        out[e] = (uint16_t)v;
      }
    }
    wave_sync();                                             /* codes[] is reused by the wave's next block */
  }
}

/* ------------------------------------------------------------------------------------------------------------------
 * k_intra_plan: PLAN_SPLIT workgroups of 4 waves per CTB with intra blocks; a wave takes every (4 * PLAN_SPLIT)-th block, one
 * border entry per lane (up to three passes for the 129 entries of a 32x32 block).  (Pictures with a handful of intra blocks per CTB;
 * an intra picture's CTBs are planned by k_intra itself.)
 * ---------------------------------------------------------------------------------------------------------------- */
template <int CF>
__device__ __forceinline__ void k_intra_plan_body(const DevPic& p, int work_n, const int item, const int part, const int n_parts)
{
  M355_GATE(p);
  __shared__ uint16_t s_code[4][4 * 32 + 8];
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), nwv = (int)(blockDim.x >> 6);   /* (4 waves per workgroup, or 1: k_residual_tu_plan) */
  if (item >= work_n) return;
  const DevIntraWork* wp = p.intra_work + item;
  const int ctb = __builtin_amdgcn_readfirstlane((int)wp->ctb);
  const uint32_t ib_start = (uint32_t)__builtin_amdgcn_readfirstlane((int)wp->ib_start), ib_count = (uint32_t)__builtin_amdgcn_readfirstlane((int)wp->ib_count);
  const uint32_t nb_same = (uint32_t)__builtin_amdgcn_readfirstlane((int)wp->nb_same), nb_earlier = (uint32_t)__builtin_amdgcn_readfirstlane((int)wp->nb_earlier);
  const uint32_t plan_base = (uint32_t)__builtin_amdgcn_readfirstlane((int)wp->plan_base);
  const int ctbX = ctb % p.ctbW, ctbY = ctb / p.ctbW;
  for (uint32_t k = (uint32_t)(wv + nwv * part); k < ib_count; k += (uint32_t)(nwv * n_parts))
    d_intra_plan_block<CF>(p, ib_start + k, ctbX, ctbY, nb_same, nb_earlier, s_code[wv], p.iplan + plan_base);
}

#endif
