/*
 * k_intra.hip — intra prediction + deferred residual add, as a dependency-driven CTB wavefront.
 *
 * Replaces decode_intra_prediction (intrapred.cc:277-369): intra_border_computer::preproc /
 * fill_from_image / reference_sample_substitution (intrapred.h:436-674),
 * intra_prediction_sample_filtering (intrapred.h:185-258), intra_prediction_planar / _DC / _angular
 * (intrapred.h:261-433), the residual add of decode_TU (slice.cc:3460-3524) for intra blocks, and
 * read_pcm_samples (slice.cc:4211-4255) as a raw block.
 *
 * Why it looks like this: intra prediction of a block reads the reconstructed samples of its
 * left / above / above-right neighbours, so the blocks of one CTB form a serial chain and CTBs form
 * the classic WPP wavefront (needs CTB (x+1,y-1)).  This is the reference's ctb_progress protocol
 * (image.h:76-80, slice.cc:4789-4795) moved onto the device:
 *   - one workgroup per CTB that contains intra blocks, described by one host-prepared 32-byte record (DevIntraWork: block
 *     range, wave count, the 3x3 neighbourhood's slice / tile / decode-order facts).  CTBs that read no intra sample of a
 *     neighbour are taken by workgroup index; the dependent ones are claimed from an atomic ticket in DECODE (tile-scan)
 *     order, so a workgroup only ever waits on CTBs claimed before it (or on ticket-free ones, which wait for nobody and
 *     are dispatched first);
 *   - dependencies between CTBs are tracked at the granularity of the SAMPLES ACTUALLY READ, not per CTB: a block that
 *     finishes a piece of its CTB's right column or bottom row publishes those samples as 8-byte granules
 *     {tag = this decode's epoch, two samples} with agent-scope (write-through) stores — "the data is the flag"
 *     (MI355X guide, guideline 16, form R2: no fences, no write-back of the L2).  A consumer stages its halo from the
 *     picture where the neighbouring samples come from the preceding kernels (inter prediction: final by stream order)
 *     and from the granules where they come from an intra block of a neighbour CTB; a granule that is not there yet is
 *     polled only by the block whose border gather needs it, when it needs it.  A CTB therefore starts at once, runs
 *     beside its neighbours, and the critical path of an intra picture is the diagonal of BLOCKS (about 15 levels of
 *     lag per CTB column and 30 per CTB row for a picture of 4x4 blocks), not (W_ctb + 2 H_ctb) whole-CTB steps;
 *   - inside the workgroup the CTB is resident in LDS and its blocks run LEVEL BY LEVEL: the host sorts each CTB's
 *     blocks by dependency level (runtime.hip intra_schedule: a block depends on the earlier blocks that cover its left
 *     column / top row), blocks of one level are independent, and each colour component has 1, 2 or 4 wavefronts (per
 *     CTB, from the widest level: a lone intra CU in an inter picture needs one, a CTB of 4x4 blocks four) that share a
 *     level's blocks, with a workgroup barrier between levels when there is more than one wave per component — the
 *     serial chain per CTB shrinks from the block count to the level count (about 46 instead of 256 for 4x4 blocks);
 *   - the inverse transforms were done up front, in parallel, by k_residual.
 * This stage is dependency-bound, not bandwidth-bound.
 */
#include <stdlib.h>
#include "k_common.h"

#define MAXCTB 64
/* body rows of a component: CTB width + 8 samples — sample x lives at column x + 8 (16-byte aligned 8-sample vectors),
   the left halo at column 7.  Chroma bodies are sized for the chroma format (template parameter CF): the LDS
   footprint decides how many CTBs a CU works on at once, and this stage lives on concurrency. */
#define BODY_PITCH_OF(cw) ((cw) + 8)
#define BODY_X0 8
#define SPIN_LIMIT M355_SPIN_LIMIT   /* k_asm.h: bound on the polls for one granule (a list that promises a sample nobody produces) */

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

#define HALO_NOT_READY 0xFFFFFFFFu

typedef unsigned long long m355_granule;   /* (epoch << 32) | sample1 << 16 | sample0 */

/* right-column granules of CTB column `col` of component c: index (row of the picture) >> 1 */
__device__ __forceinline__ m355_granule* d_edge_col(const DevPic& p, int c, int col, int y) { return p.edge + p.edge_col_ofs[c] + (size_t)col * (size_t)(p.ph[c] >> 1) + (size_t)(y >> 1); }
/* bottom-row granules of CTB row `row`: index (column of the picture) >> 1 */
__device__ __forceinline__ m355_granule* d_edge_row(const DevPic& p, int c, int row, int x) { return p.edge + p.edge_row_ofs[c] + (size_t)row * (size_t)(p.pw[c] >> 1) + (size_t)(x >> 1); }

/* NW = waves per workgroup: 12 for intra pictures (CTBs with hundreds of blocks: up to 8 luma + 2 + 2 chroma waves share a
 * level, the CTB's residuals are fetched into LDS up front), 4 for inter pictures (a handful of intra blocks per CTB: up to
 * 1-2 + 1 + 1 waves, residual cache lines requested in the prologue and loaded per block behind its border gather; the
 * smaller footprint keeps ~2.5x as many CTBs in flight).  The CTB's own wave counts come from DevIntraWork.waves_code
 * (runtime.hip intra_schedule). */
template <class PIX, int CF, int NW>
__global__ void __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(4))) k_intra(DevPic p, int work_n)
{
  M355_GATE(p);
  constexpr bool RES_LDS = NW >= 12;
  /* per component: halo (top row x = -1 .. 2*cw-1 at index x+1; left column) as 32-bit words: a sample, or
     HALO_NOT_READY while the neighbour CTB has not published it; body rows */
  constexpr int CW_C = (CF == 1 || CF == 2) ? MAXCTB / 2 : MAXCTB, CH_C = CF == 1 ? MAXCTB / 2 : MAXCTB;   /* chroma CTB size */
  constexpr int BODY_L = MAXCTB * BODY_PITCH_OF(MAXCTB), BODY_C = CF == 0 ? 8 : CH_C * BODY_PITCH_OF(CW_C);
  __shared__ uint32_t s_top[3][2 * MAXCTB + 2];
  __shared__ uint32_t s_left[3][MAXCTB];
  __shared__ __attribute__((aligned(16))) uint16_t s_body[BODY_L + 2 * BODY_C];
  /* the CTB's deferred residuals in picture layout (pitch = component CTB width): fetched up front, all loads in flight
     together, so that the per-block chain reads them from LDS instead of paying a global-memory latency per block */
  constexpr int RES_L = RES_LDS ? MAXCTB * MAXCTB : 8, RES_C = (CF == 0 || !RES_LDS) ? 8 : CH_C * CW_C;
  __shared__ __attribute__((aligned(16))) int16_t s_res[RES_L + 2 * RES_C];
  /* per WAVE (a wave works on one block at a time): */
  constexpr int NWV = NW;
  __shared__ uint16_t s_raw[NWV][4 * 32 + 8];      /* gathered border, entry e = i + 2nT */
  __shared__ uint16_t s_p[NWV][4 * 32 + 8];        /* substituted border */
  __shared__ uint16_t s_f[NWV][4 * 32 + 8];        /* filtered border */
  __shared__ uint32_t s_ticket;
  __shared__ uint32_t s_need[3][MAXCTB];         /* per component and CTB row, which 8-sample vectors some block's border reads */
  __shared__ uint32_t s_touch[64];               /* d_touch scratch (never read) */
  __shared__ uint32_t s_hneed[3][8];             /* ... and which halo entries (bit h: top entries 0 .. 2cw, then the left column) */

  /* everything derived from the wave index or from a block record is wave-uniform: say so (readfirstlane / readlane), so that
     the component's plane pointers, pitches and granule offsets are scalar loads from the kernel arguments instead of vector
     loads (each one a wait on the in-order vector-memory counter, i.e. on every sample store still in flight), record
     fields live in SGPRs and the per-block branches are scalar */
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;

  /* work item: the first n_intra_free items are CTBs that wait for no neighbour — any workgroup may take any of them, so they go
     by workgroup index (no atomic, no barrier); the dependent CTBs behind them are claimed through the ticket, in decode order,
     so that a workgroup only ever waits on items claimed before its own (free ones, or lower tickets) */
  int item = (int)blockIdx.x;
  if (item >= p.n_intra_free) {          /* (uniform per workgroup) */
    if (threadIdx.x == 0) s_ticket = (uint32_t)p.n_intra_free + atomicAdd(p.ticket, 1u);
    __syncthreads();
    item = __builtin_amdgcn_readfirstlane((int)s_ticket);
  }
  if (item >= work_n) return;
  /* the CTB's descriptor (host-prepared, runtime.hip): one scalar 32-byte load */
  const DevIntraWork* wp = p.intra_work + item;
  const uint4 wd0 = *(const uint4*)wp;
  const uint32_t wd1 = ((const uint32_t*)wp)[4];
  const int ctb = __builtin_amdgcn_readfirstlane((int)wd0.x);
  struct { uint32_t ib_start, ib_count; } ctbinfo = {(uint32_t)__builtin_amdgcn_readfirstlane((int)wd0.y), (uint32_t)__builtin_amdgcn_readfirstlane((int)wd0.z)};
  const uint32_t nb_same = (uint32_t)__builtin_amdgcn_readfirstlane((int)(wd0.w & 0xFFFFu)), nb_earlier = (uint32_t)__builtin_amdgcn_readfirstlane((int)(wd0.w >> 16));
  /* wave -> (colour component c, sub-wave g of G): GL luma waves, then GC for Cb, GC for Cr; the rest leave at once
     (finished waves do not take part in later barriers) */
  int GL, GC;
  {
    const int code = __builtin_amdgcn_readfirstlane((int)(wd1 & 3u));   /* widest level of the CTB: luma 1 / 2 / 3-4 / more blocks */
    GL = code == 0 ? 1 : (code == 1 ? 2 : (code == 2 || NW < 12 ? 4 : 8));
    GC = (code == 3 && NW >= 12) ? 2 : 1;
    if (GL > NW - 2) GL = NW - 2;
  }
  if (wv >= GL + 2 * GC) return;
  const int c = wv < GL ? 0 : (wv < GL + GC ? 1 : 2);
  const int G = c == 0 ? GL : GC, g = c == 0 ? wv : (wv - GL - (c - 1) * GC);
  const bool multi = GL + GC > 2;                        /* more than one wave per component somewhere: levels end in a barrier */
  const int ctbX = ctb % p.ctbW, ctbY = ctb / p.ctbW;
  const int l2c = p.pp.log2_ctb_size;

  const int nc = p.pp.chroma_format_idc ? 3 : 1;
  const bool comp = c < nc;
  const int csw = c ? (p.sw == 2) : 0, csh = c ? (p.sh == 2) : 0;
  const int SubW = 1 << csw, SubH = 1 << csh;
  const int cw = (1 << l2c) >> csw, ch = (1 << l2c) >> csh;
  const int x0c = (ctbX << l2c) >> csw, y0c = (ctbY << l2c) >> csh;
  const int bd = c ? p.pp.bit_depth_chroma : p.pp.bit_depth_luma;
  const int cs = comp ? c : 0;
  PIX* plane = (PIX*)p.plane[cs];
  const int stride = p.stride[cs], pw = p.pw[cs], ph = p.ph[cs];
  uint32_t* top = s_top[cs];
  uint32_t* left = s_left[cs];
  uint16_t* body = s_body + (cs == 0 ? 0 : BODY_L + (cs - 1) * BODY_C);
  const int BODY_PITCH = cs == 0 ? BODY_PITCH_OF(MAXCTB) : BODY_PITCH_OF(CW_C);
  uint16_t* raw = s_raw[wv];
  uint16_t* psub = s_p[wv];
  uint16_t* pf = s_f[wv];
  const uint32_t epoch = p.epoch;
#define SYNC_CTB() do { if (multi) __syncthreads(); else wave_sync(); } while (0)
  int16_t* resl = s_res + (cs == 0 ? 0 : RES_L + (cs - 1) * RES_C);
  const int RES_PITCH = cs == 0 ? MAXCTB : CW_C;

  /* ---- which body vectors does some block's border read?  Only those are staged: the row above a block
     (x-1 .. x+2nT-1) and the column left of it (y .. y+2nT-1), intrapred.h:436-674 — a 64x64 CTB with two 8x8 intra
     blocks: ~6 of its 512 luma vectors.  Samples that blocks of this CTB produce land in the same LDS tile later. ---- */
  if (comp && g == 0) {
    for (int y = lane; y < ch; y += 64) s_need[cs][y] = 0;
    if (lane < 8) s_hneed[cs][lane] = 0;
    wave_sync();
    const int nvr = cw >> 3;                                /* vectors per row */
    for (uint32_t k = lane; k < ctbinfo.ib_count; k += 64) {
      const uint32_t* r = (const uint32_t*)&p.ibs[ctbinfo.ib_start + k];
      const uint32_t w0 = r[0], w1 = r[1];
      if ((w1 & 0xFFu) != (uint32_t)c) continue;
      const int nT = 1 << ((w1 >> 8) & 0xFFu);
      const int lx = (int)(w0 & 0xFFFFu) - x0c, ly = (int)(w0 >> 16) - y0c;
      if (!RES_LDS && ((w1 >> 24) & M355_IBF_HAS_RESIDUAL) && !((w1 >> 24) & M355_IBF_PCM)) {
        /* the block's residual is read when its dependency level comes up: ask for its cache lines now */
        const char* rp_ = (const char*)(p.resbuf + r[2]);
        for (int o = 0; o < nT * nT * 2; o += 128) d_touch(rp_ + o, s_touch);
      }
      if (ly >= 1) {
        const int v0 = max(lx - 1, 0) >> 3, v1 = min((lx + 2 * nT - 1) >> 3, nvr - 1);
        if (v1 >= v0) atomicOr(&s_need[cs][ly - 1], ((2u << v1) - 1u) & ~((1u << v0) - 1u));
      }
      if (lx >= 1) {
        const uint32_t bit = 1u << ((lx - 1) >> 3);
        const int y1 = min(ly + 2 * nT, ch);
        for (int y = max(ly, 0); y < y1; y++) atomicOr(&s_need[cs][y], bit);
      }
      /* halo entries behind this block's border: top[lx .. lx+2nT] for a block in the CTB's first row, left[ly-1 .. ly+2nT-1]
         for one in its first column (ranges as 32-bit word masks) */
      auto need_range = [&](int h0, int h1) {              /* halo entries h0 .. h1 inclusive */
        for (int w = h0 >> 5; w <= (h1 >> 5); w++) {
          const int a = max(h0 - 32 * w, 0), b = min(h1 - 32 * w, 31);
          atomicOr(&s_hneed[cs][w], (b >= 31 ? ~0u : ((2u << b) - 1u)) & ~((1u << a) - 1u));
        }
      };
      if (ly == 0) need_range(lx, min(lx + 2 * nT, 2 * cw));
      if (lx == 0) need_range(2 * cw + 1 + max(ly - 1, 0), 2 * cw + 1 + min(ly + 2 * nT - 1, ch - 1));
    }
  }
  SYNC_CTB();
  if (comp) {
    /* ---- stage those vectors: 8 samples each; four per lane are requested before the first is stored (a loop of
       load -> store steps costs one memory round trip per step, and a CTB has up to eight steps per lane) ---- */
    {
      const int l2v = (l2c - csw) - 3;                       /* log2(vectors per row); cw >= 8 */
      const int nvec = ch << l2v;
      constexpr int U = 4;
      for (int idx0 = lane + 64 * g; idx0 < nvec; idx0 += 64 * G * U) {
        uint4 v[U];
        bool take[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
          const int idx = idx0 + u * 64 * G;
          const int y = idx >> l2v, xv = (idx & ((1 << l2v) - 1)) * 8;
          take[u] = idx < nvec && ((s_need[cs][min(y, ch - 1)] >> (xv >> 3)) & 1u) && x0c + xv < pw && y0c + y < ph;
          v[u] = make_uint4(0, 0, 0, 0);
          if (take[u]) {
            const PIX* src = plane + (size_t)(y0c + y) * stride + x0c + xv;
            if (sizeof(PIX) == 2) v[u] = *(const uint4*)src;
            else { const uint2 b = *(const uint2*)src; v[u].x = b.x; v[u].y = b.y; }
          }
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
          if (!take[u]) continue;
          const int idx = idx0 + u * 64 * G;
          const int y = idx >> l2v, xv = (idx & ((1 << l2v) - 1)) * 8;
          uint4 o = v[u];
          if (sizeof(PIX) == 1) {
            const uint32_t bx = v[u].x, by = v[u].y;
            o.x = (bx & 0xFFu) | ((bx & 0xFF00u) << 8); o.y = ((bx >> 16) & 0xFFu) | ((bx >> 8) & 0xFF0000u);
            o.z = (by & 0xFFu) | ((by & 0xFF00u) << 8); o.w = ((by >> 16) & 0xFFu) | ((by >> 8) & 0xFF0000u);
          }
          *(uint4*)(body + y * BODY_PITCH + BODY_X0 + xv) = o;
        }
      }
    }
    /* ---- halo: the row above the CTB (x = -1 .. 2cw-1) and the column left of it.  A sample that an INTRA block of a
       neighbour CTB produces comes from that CTB's granules — if it is there already; otherwise the entry stays
       HALO_NOT_READY and the block that needs it polls for it.  Everything else was finished by the preceding kernels
       (stream order) and is read from the picture. ---- */
    const int nhalo = (2 * cw + 1) + ch;
    /* only the entries some block's border reads (s_hneed); per entry up to three dependent loads (CU plane -> CU record ->
       sample or granule): the lane's (up to four) entries go through each step together */
    {
      constexpr int U = 4;
      for (int h0 = lane + 64 * g; h0 < nhalo; h0 += 64 * G * U) {
        int hx[U], hy[U];
        bool in[U], top_[U];
        uint32_t ci[U], val[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
          const int h = h0 + u * 64 * G;
          top_[u] = h < 2 * cw + 1;
          hx[u] = top_[u] ? x0c - 1 + h : x0c - 1; hy[u] = top_[u] ? y0c - 1 : y0c + (h - (2 * cw + 1));
          in[u] = h < nhalo && ((s_hneed[cs][min(h, nhalo - 1) >> 5] >> (h & 31)) & 1u) && hx[u] >= 0 && hy[u] >= 0 && hx[u] < pw && hy[u] < ph;
          ci[u] = in[u] ? d_cu_index_at(p, hx[u] << csw, hy[u] << csh) : 0u;
        }
        bool intra[U];
#pragma unroll
        for (int u = 0; u < U; u++) intra[u] = in[u] && ci[u] != 0 && p.cus[ci[u] - 1].pred_mode == 0;
#pragma unroll
        for (int u = 0; u < U; u++) {
          val[u] = 0;
          if (intra[u]) {
            const m355_granule gr = __hip_atomic_load(top_[u] ? d_edge_row(p, cs, ctbY - 1, hx[u]) : d_edge_col(p, cs, ctbX - 1, hy[u]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            val[u] = (uint32_t)(gr >> 32) == epoch ? (uint32_t)((gr >> (16 * ((top_[u] ? hx[u] : hy[u]) & 1))) & 0xFFFFu) : HALO_NOT_READY;
          } else if (in[u]) val[u] = plane[(size_t)hy[u] * stride + hx[u]];
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
          const int h = h0 + u * 64 * G;
          if (h >= nhalo) continue;
          if (top_[u]) top[h] = val[u]; else left[h - (2 * cw + 1)] = val[u];
        }
      }
    }
  }
  /* ---- residual pre-pass (intra pictures): the CTB's deferred residuals go to LDS, each component's waves taking its
     blocks in turn ---- */
  if (RES_LDS) {
    int taken = 0;                                           /* blocks of this component seen so far (wave-uniform) */
    for (uint32_t kbase = 0; kbase < ctbinfo.ib_count; kbase += 64) {
      uint32_t rw0 = 0, rw1 = 0xFF, rw2 = 0;
      if (kbase + lane < ctbinfo.ib_count) {
        const uint32_t* r = (const uint32_t*)&p.ibs[ctbinfo.ib_start + kbase + lane];
        rw0 = r[0]; rw1 = r[1]; rw2 = r[2];
      }
      unsigned long long mine = __ballot((int)(comp && (rw1 & 0xFFu) == (uint32_t)c));
      while (mine) {
        const int src = __ffsll(mine) - 1;
        mine &= mine - 1;
        if ((taken++ & (G - 1)) != g) continue;
        const uint32_t w0 = __builtin_amdgcn_readlane(rw0, src), w1 = __builtin_amdgcn_readlane(rw1, src), w2 = __builtin_amdgcn_readlane(rw2, src);
        const int flags = (int)(w1 >> 24), log2 = (int)((w1 >> 8) & 0xFFu), nT = 1 << log2;
        if (!(flags & M355_IBF_HAS_RESIDUAL) || (flags & M355_IBF_PCM)) continue;
        const int lx = (int)(w0 & 0xFFFFu) - x0c, ly = (int)(w0 >> 16) - y0c;
        for (int o = lane * 4; o < nT * nT; o += 256) {        /* 4 samples (one row segment) per lane and step */
          const uint2 v = *(const uint2*)(p.resbuf + w2 + o);
          *(uint2*)(resl + (ly + (o >> log2)) * RES_PITCH + lx + (o & (nT - 1))) = v;
        }
      }
    }
  }
  __syncthreads();     /* bodies, halos and residuals staged */
  /* The CTB's block records (sorted by level, then component) are fetched 64 at a time (one per lane, coalesced) by
     EVERY wave; for each level present in the batch, a wave takes the blocks of its component that fall to it
     (every G-th) by broadcasting the record from the owning lane, then all waves meet at the workgroup barrier.  The
     loop bounds come from the records alone, so all waves (also those of absent components) execute the same
     barriers. */
  for (uint32_t kbase = 0; kbase < ctbinfo.ib_count; kbase += 64) {
    uint32_t rw0 = 0, rw1 = 0xFF, rw2 = 0;
    int lv = -1;
    const int nvalid = min(64, (int)(ctbinfo.ib_count - kbase));
    if (lane < nvalid) {
      const uint32_t* r = (const uint32_t*)&p.ibs[ctbinfo.ib_start + kbase + lane];
      rw0 = r[0]; rw1 = r[1]; rw2 = r[2];
      lv = p.ib_level[ctbinfo.ib_start + kbase + lane];
    }
    const int lv_first = __builtin_amdgcn_readlane(lv, 0), lv_last = __builtin_amdgcn_readlane(lv, nvalid - 1);
    for (int L = lv_first; L <= lv_last; L++) {
    unsigned long long mine = __ballot((int)(comp && lv == L && (rw1 & 0xFFu) == (uint32_t)c));
    int rank = 0;
    while (mine) {
      const int src = __ffsll(mine) - 1;
      mine &= mine - 1;
      if ((rank++ & (G - 1)) != g) continue;             /* another wave of this component takes it */
      m355_ib ib;
      {
        const uint32_t w0 = __builtin_amdgcn_readlane(rw0, src), w1 = __builtin_amdgcn_readlane(rw1, src), w2 = __builtin_amdgcn_readlane(rw2, src);
        ib.x = (uint16_t)(w0 & 0xFFFFu); ib.y = (uint16_t)(w0 >> 16);
        ib.cidx = (uint8_t)(w1 & 0xFFu); ib.log2_size = (uint8_t)((w1 >> 8) & 0xFFu); ib.mode = (uint8_t)((w1 >> 16) & 0xFFu); ib.flags = (uint8_t)(w1 >> 24);
        ib.res_ofs = w2;
      }
      const int nT = 1 << ib.log2_size;
      const int xB = ib.x, yB = ib.y, lx = xB - x0c, ly = yB - y0c;
      /* does the block complete a piece of the CTB's right column / bottom row that a neighbour CTB may read? */
      const bool pub_col = lx + nT == cw && ctbX + 1 < p.ctbW, pub_row = ly + nT == ch && ctbY + 1 < p.ctbH;

      if (!(ib.flags & M355_IBF_PCM)) {
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

      /* ---- fill_from_image (intrapred.h:534-633): one border entry per lane and chunk ---- */
      unsigned long long am[3] = {0, 0, 0};
#pragma unroll
      for (int q = 0; q < 3; q++) {
        if (64 * q >= nEnt) continue;              /* wave-uniform: 4x4 / 8x8 blocks have 17 / 33 border entries */
        const int e = lane + 64 * q;
        bool av = false;
        uint32_t val = 0;
        uint32_t* hslot = nullptr;                 /* the halo word behind this entry, if it is one */
        const m355_granule* gsrc = nullptr;
        int ghalf = 0;
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
          if (av) {
            if (sy < 0) { hslot = &top[sx + 1]; val = *hslot; gsrc = d_edge_row(p, cs, ctbY - 1, x0c + sx); ghalf = (x0c + sx) & 1; }
            else if (sx < 0) { hslot = &left[sy]; val = *hslot; gsrc = d_edge_col(p, cs, ctbX - 1, y0c + sy); ghalf = (y0c + sy) & 1; }
            else val = body[sy * BODY_PITCH + sx + BODY_X0];
          }
        }
        /* a halo sample its CTB has not published yet: poll its granule (relaxed, agent scope: served by the L2 / fabric,
           never by this CU's L1); every waiting lane has its own word, the wave leaves when all have arrived */
        bool pending = av && hslot != nullptr && val == HALO_NOT_READY;
        if (__any(pending)) {
          unsigned spins = 0;
          for (;;) {
            if (pending) {
              const m355_granule gr = __hip_atomic_load(gsrc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              if ((uint32_t)(gr >> 32) == epoch) { val = (uint32_t)((gr >> (16 * ghalf)) & 0xFFFFu); *hslot = val; pending = false; }
            }
            if (!__any(pending)) break;
            __builtin_amdgcn_s_sleep(1);
            if (++spins > SPIN_LIMIT) { if (lane == 0) atomicExch(p.timeout, 1u); break; }
          }
        }
        if (e < nEnt) raw[e] = (uint16_t)val;
        am[q] = __ballot(av);
      }
      wave_sync();
      /* residual of this block (written by k_residual; its cache lines were requested in the prologue): the loads are issued
         here, BEHIND the border gather — hipcc drains the vector-memory counter in front of the gather's poll loop, so loads
         issued before it are waited for at once — and consumed after substitution / smoothing, up to 16 samples per lane (32x32) */
      int16_t rv[16];
#pragma unroll
      for (int q = 0; q < 16; q++) {
        const int o = lane + 64 * q;
        rv[q] = (!RES_LDS && (ib.flags & M355_IBF_HAS_RESIDUAL) && o < nT * nT) ? p.resbuf[ib.res_ofs + o] : (int16_t)0;
      }
      /* ---- reference_sample_substitution (only when something is missing: the common interior block keeps its gathered
         border as it is) ---- */
      const bool none = (am[0] | am[1] | am[2]) == 0;
      bool all_av;
      {
        const unsigned long long full = ~0ull;
        const int r1 = nEnt - 64, r2 = nEnt - 128;      /* entries in the 2nd / 3rd mask word */
        all_av = am[0] == (nEnt >= 64 ? full : ((1ull << nEnt) - 1ull)) &&
                 (r1 <= 0 || am[1] == (r1 >= 64 ? full : ((1ull << r1) - 1ull))) &&
                 (r2 <= 0 || am[2] == ((1ull << r2) - 1ull));
      }
      uint16_t* pp_ = raw;
      if (!all_av) {
#pragma unroll
        for (int q = 0; q < 3; q++) {
          if (64 * q >= nEnt) continue;
          const int e = lane + 64 * q;
          if (e < nEnt) {
            int v;
            if (none) v = 1 << (bd - 1);
            else if ((am[q] >> lane) & 1) v = raw[e];
            else v = raw[d_subst_src(e, am[0], am[1], am[2])];
            psub[e] = (uint16_t)v;
          }
        }
        wave_sync();
        pp_ = psub;
      }
      /* ---- intra_prediction_sample_filtering (intrapred.h:185-258) ---- */
      const int mode = ib.mode;
      uint16_t* P = pp_; /* border in use, entry index = i + 2nT */
      const int Z = 2 * nT;
      if (!(p.pp.flags & M355_PF_INTRA_SMOOTHING_DISABLED) && (c == 0 || p.pp.chroma_format_idc == 3) && mode != 1 && nT != 4) {
        const int minDist = min(d_abs(mode - 26), d_abs(mode - 10));
        const bool filt = nT == 8 ? minDist > 7 : (nT == 16 ? minDist > 1 : (nT == 32 ? minDist > 0 : false));
        if (filt) {
          const bool bi = (p.pp.flags & M355_PF_STRONG_INTRA_SMOOTHING) && c == 0 && nT == 32 &&
                          d_abs((int)pp_[Z] + pp_[Z + 64] - 2 * pp_[Z + 32]) < (1 << (p.pp.bit_depth_luma - 5)) &&
                          d_abs((int)pp_[Z] + pp_[Z - 64] - 2 * pp_[Z - 32]) < (1 << (p.pp.bit_depth_luma - 5));
#pragma unroll
          for (int q = 0; q < 3; q++) {
            if (64 * q >= nEnt) continue;
            const int e = lane + 64 * q;
            if (e < nEnt) {
              const int i = e - Z;
              int v;
              if (i == -Z || i == Z) v = pp_[e];
              else if (bi) {
                if (i == 0) v = pp_[Z];
                else if (i < 0) v = pp_[Z] + (((-i) * ((int)pp_[Z - 64] - pp_[Z]) + 32) >> 6);
                else v = pp_[Z] + ((i * ((int)pp_[Z + 64] - pp_[Z]) + 32) >> 6);
              } else v = (pp_[e + 1] + 2 * pp_[e] + pp_[e - 1] + 2) >> 2;
              pf[e] = (uint16_t)v;
            }
          }
          wave_sync();
          P = pf;
        }
      }
#define BRD(i) ((int)P[(i) + Z])
      /* ---- prediction (intrapred.h:261-433) ---- */
      const int log2 = ib.log2_size;
      int dcVal = 0;
      if (mode == 1) {
        int s = 0;
        if (lane < nT) s = BRD(lane + 1) + BRD(-lane - 1);
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
        dcVal = (s + nT) >> (log2 + 1);
      }
      /* angular modes (intrapred.h:330-433): the projected reference array ref[] of the reference is not built — its entry x is
         border entry sgn*x for x >= 0 and, left of the corner (negative angles only), -sgn*((x*invAngle+128)>>8): the two taps
         of a sample are read straight from the border */
      /* intraPredAngle / invAngle (intrapred.h:313-326) from the distance d of the mode to the pure horizontal (10) / vertical
         (26) mode, looked up in packed constants: a table in memory would be a dependent vector load per block (the mode is
         per-lane data to the compiler), i.e. a memory round trip on the block chain — and, the vector-memory counter being
         in-order, a wait for everything else this wave has in flight */
      const int d_ang = mode >= 18 ? d_abs(mode - 26) : d_abs(mode - 10);                    /* 0..8 */
      const int mag = (int)((0x20345488D1214100ull >> (7 * d_ang)) & 0x7Full);                 /* {0,2,5,9,13,17,21,26,32}, 7 bits each */
      const bool neg = mode >= 18 ? mode < 26 : mode > 10;
      const int angle = mode < 2 ? 0 : (neg ? -mag : mag);
      const int sgn = mode >= 18 ? 1 : -1;
      /* invAngle = -round(8192 / |angle|): {4096,1638,910,630 | 482,390,315,256} for d = 1..8, 16 bits each */
      const unsigned long long inv_tab = d_ang <= 4 ? 0x0276038E06661000ull : 0x0100013B018601E2ull;
      const int inv = (mode >= 2 && angle < 0) ? -(int)((inv_tab >> (16 * ((d_ang - 1) & 3))) & 0xFFFFull) : 0;
#define REFV(x_) ((x_) >= 0 ? BRD(sgn * (x_)) : BRD(-sgn * (((x_) * inv + 128) >> 8)))
      const bool has_res = (ib.flags & M355_IBF_HAS_RESIDUAL) != 0;
      const bool edge = (c == 0 && nT < 32);
      const bool bfilt = edge && !(ib.flags & M355_IBF_DISABLE_BOUNDARY_FILTER);
#pragma unroll
      for (int q = 0; q < 16; q++) {
        const int o = lane + 64 * q;
        if (o >= nT * nT) break;
        const int y = o >> log2, x = o & (nT - 1);
        int v;
        if (mode == 0) {
          v = ((nT - 1 - x) * BRD(-1 - y) + (x + 1) * BRD(1 + nT) + (nT - 1 - y) * BRD(1 + x) + (y + 1) * BRD(-1 - nT) + nT) >> (log2 + 1);
        } else if (mode == 1) {
          v = dcVal;
          if (edge) {
            if (x == 0 && y == 0) v = (BRD(-1) + 2 * dcVal + BRD(1) + 2) >> 2;
            else if (y == 0) v = (BRD(x + 1) + 3 * dcVal + 2) >> 2;
            else if (x == 0) v = (BRD(-y - 1) + 3 * dcVal + 2) >> 2;
          }
        } else {
          const int a = mode >= 18 ? y : x, b = mode >= 18 ? x : y;
          const int iIdx = ((a + 1) * angle) >> 5, iFact = ((a + 1) * angle) & 31;
          const int r1 = REFV(b + iIdx + 1);
          v = iFact ? ((32 - iFact) * r1 + iFact * REFV(b + iIdx + 2) + 16) >> 5 : r1;
          if (bfilt) {
            if (mode == 26 && x == 0) v = d_clip_bd(BRD(1) + ((BRD(-1 - y) - BRD(0)) >> 1), bd);
            if (mode == 10 && y == 0) v = d_clip_bd(BRD(-1) + ((BRD(1 + x) - BRD(0)) >> 1), bd);
          }
        }
        if (has_res) v = d_clip_bd(v + (RES_LDS ? (int)resl[(ly + y) * RES_PITCH + lx + x] : (int)rv[q]), bd);
        body[(ly + y) * BODY_PITCH + lx + x + BODY_X0] = (uint16_t)v;     /* for the next blocks' borders */
        plane[(size_t)(yB + y) * stride + xB + x] = (PIX)v;                /* the picture: only intra samples are (re)written */
      }
#undef BRD
#undef REFV
      } else { /* raw block (slice.cc:4211-4255) */
        for (int o = lane; o < nT * nT; o += 64) {
          const int y = o >> ib.log2_size, x = o & (nT - 1);
          const uint16_t v = p.pcm[ib.res_ofs + o];
          body[(ly + y) * BODY_PITCH + lx + x + BODY_X0] = v;
          plane[(size_t)(yB + y) * stride + xB + x] = (PIX)v;
        }
      }
      wave_sync();
      /* ---- publish: the block's share of the CTB's right column / bottom row, two samples per granule ---- */
      if (pub_col && lane < (nT >> 1)) {
        const int y = ly + 2 * lane;
        const uint32_t s0 = body[y * BODY_PITCH + lx + nT - 1 + BODY_X0], s1 = body[(y + 1) * BODY_PITCH + lx + nT - 1 + BODY_X0];
        __hip_atomic_store(d_edge_col(p, cs, ctbX, y0c + y), ((m355_granule)epoch << 32) | (s1 << 16) | s0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (pub_row && lane >= 32 && lane < 32 + (nT >> 1)) {
        const int x = lx + 2 * (lane - 32);
        const uint32_t s0 = body[(ly + nT - 1) * BODY_PITCH + x + BODY_X0], s1 = body[(ly + nT - 1) * BODY_PITCH + x + 1 + BODY_X0];
        __hip_atomic_store(d_edge_row(p, cs, ctbY, x0c + x), ((m355_granule)epoch << 32) | (s1 << 16) | s0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }   /* this wave's blocks of the level */
    if (multi) __syncthreads();
    /* level done: its samples are in LDS for the next level's borders (one wave per component: its own blocks are ordered
       by wave_sync above; components do not interact) */
    }   /* levels in the batch */
  }   /* 64-record batches */
#undef SYNC_CTB
}

template <class PIX, int CF>
static void launch_intra_cf(const DevPic& p, hipStream_t st)
{
  hipMemsetAsync(p.ticket, 0, 4, st);
  /* dense intra pictures: 12 waves (up to 8 luma + 2 + 2 chroma blocks of a level at once); sparse ones: 4 (3 and 6 measured
     slower) */
  if (p.intra_dense) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_intra<PIX, CF, 12>), dim3(p.n_intra_work), dim3(64 * 12), 0, st, p, p.n_intra_work);
  else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_intra<PIX, CF, 4>), dim3(p.n_intra_work), dim3(64 * 4), 0, st, p, p.n_intra_work);
}

void m355_launch_intra(const DevPic& p, bool hbd, hipStream_t st)
{
  if (!p.n_intra_work) return;
  switch (p.pp.chroma_format_idc) {
    case 0: if (hbd) launch_intra_cf<uint16_t, 0>(p, st); else launch_intra_cf<uint8_t, 0>(p, st); break;
    case 1: if (hbd) launch_intra_cf<uint16_t, 1>(p, st); else launch_intra_cf<uint8_t, 1>(p, st); break;
    case 2: if (hbd) launch_intra_cf<uint16_t, 2>(p, st); else launch_intra_cf<uint8_t, 2>(p, st); break;
    default: if (hbd) launch_intra_cf<uint16_t, 3>(p, st); else launch_intra_cf<uint8_t, 3>(p, st); break;
  }
}
