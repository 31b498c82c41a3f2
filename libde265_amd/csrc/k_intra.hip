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
 *   - one workgroup per CTB that contains intra blocks; CTBs are claimed from an atomic ticket in
 *     DECODE (tile-scan) order, so a workgroup only ever waits on CTBs claimed before it — no
 *     residency assumption, no deadlock;
 *   - completion is published per CTB with an agent-scope release + flag; consumers poll relaxed,
 *     then take ONE agent-scope acquire (MI355X guide, guideline 16);
 *   - inside the workgroup the CTB (plus top-row / left-column halo) is resident in LDS and its blocks run LEVEL BY
 *     LEVEL: the host sorts each CTB's blocks by dependency level (runtime.hip intra_schedule: a block depends on the
 *     earlier blocks that cover its left column / top row), blocks of one level are independent, and each colour
 *     component has 1 (inter pictures: a handful of blocks per CTB) or 4 (intra pictures: up to 256 luma blocks per
 *     CTB) wavefronts that share a level's blocks, with a workgroup barrier between levels — the serial chain per
 *     CTB shrinks from the block count to the level count (about 46 instead of 256 for a CTB of 4x4 blocks);
 *   - the inverse transforms were done up front, in parallel, by k_residual.
 * This stage is dependency-bound, not bandwidth-bound: critical path ~ (W_ctb + 2 H_ctb) CTB steps.
 */
#include "k_common.h"

#define MAXCTB 64
/* body rows of a component: CTB width + 8 samples — sample x lives at column x + 8 (16-byte aligned 8-sample vectors),
   the left halo at column 7.  Chroma bodies are sized for the chroma format (template parameter CF): the LDS
   footprint decides how many CTBs a CU works on at once, and this stage lives on concurrency. */
#define BODY_PITCH_OF(cw) ((cw) + 8)
#define BODY_X0 8
#define SPIN_LIMIT (1u << 24)

__constant__ int8_t c_intra_angle[35] = {0,   0,   32,  26,  21,  17, 13, 9,  5, 2, 0, -2, -5, -9, -13, -17, -21, -26,
                                         -32, -26, -21, -17, -13, -9, -5, -2, 0, 2, 5, 9,  13, 17, 21,  26,  32};
__constant__ int16_t c_intra_inv_angle[15] = {-4096, -1638, -910, -630, -482, -390, -315, -256,
                                              -315,  -390,  -482, -630, -910, -1638, -4096};

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

#define INTRA_GMAX 4   /* waves per colour component */
/* DENSE: intra pictures (many blocks per CTB): INTRA_GMAX waves per component share each level, residuals are fetched
 * into LDS up front.  !DENSE: inter pictures (a handful of blocks per CTB): one wave per component, no workgroup
 * barriers inside the chain, residuals fetched per block under its border phase. */
template <class PIX, int CF, bool DENSE>
__global__ void __launch_bounds__(DENSE ? 192 * INTRA_GMAX : 192) k_intra(DevPic p, int work_base, int work_n, int use_ticket)
{
  /* per component: top halo row (x = -1 .. 2*cw-1, index x+1) and body rows with a left halo column */
  constexpr int CW_C = (CF == 1 || CF == 2) ? MAXCTB / 2 : MAXCTB, CH_C = CF == 1 ? MAXCTB / 2 : MAXCTB;   /* chroma CTB size */
  constexpr int BODY_L = MAXCTB * BODY_PITCH_OF(MAXCTB), BODY_C = CF == 0 ? 8 : CH_C * BODY_PITCH_OF(CW_C);
  __shared__ uint16_t s_top[3][2 * MAXCTB + 2];
  __shared__ __attribute__((aligned(16))) uint16_t s_body[BODY_L + 2 * BODY_C];
  /* the CTB's deferred residuals in picture layout (pitch = component CTB width): fetched up front, all loads in flight
     together, so that the per-block chain reads them from LDS instead of paying a global-memory latency per block */
  constexpr int RES_L = DENSE ? MAXCTB * MAXCTB : 8, RES_C = (CF == 0 || !DENSE) ? 8 : CH_C * CW_C;
  __shared__ __attribute__((aligned(16))) int16_t s_res[RES_L + 2 * RES_C];
  /* per WAVE (a wave works on one block at a time): */
  constexpr int NWV = 3 * (DENSE ? INTRA_GMAX : 1);
  __shared__ uint16_t s_raw[NWV][4 * 32 + 8];      /* gathered border, entry e = i + 2nT */
  __shared__ uint16_t s_p[NWV][4 * 32 + 8];        /* substituted border */
  __shared__ uint16_t s_f[NWV][4 * 32 + 8];        /* filtered border */
  __shared__ int s_ref[NWV][3 * 32 + 8];           /* angular ref[-nT..2nT], index +32 */
  __shared__ uint32_t s_ticket;
  __shared__ uint32_t s_need[3][MAXCTB];         /* !DENSE: per component and CTB row, which 8-sample vectors some block's border reads */
  __shared__ uint32_t s_nts[9];                  /* CtbAddrRStoTS of the 3x3 CTB neighbourhood (0xFFFFFFFF outside the picture) */
  __shared__ uint8_t s_nsame[9];                 /* neighbour CTB in the picture, same slice (SliceAddrRS) and same tile */

  /* wave -> (colour component c, sub-wave g of G): blockDim.x = 192 * G */
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  constexpr int G = DENSE ? INTRA_GMAX : 1;
  const int c = wv / G, g = wv - c * G;

  if (threadIdx.x == 0) s_ticket = use_ticket ? atomicAdd(p.ticket, 1u) : blockIdx.x;
  __syncthreads();
  if ((int)s_ticket >= work_n) return;
  const int ctb = (int)p.intra_work[work_base + (int)s_ticket];
  const int ctbX = ctb % p.ctbW, ctbY = ctb / p.ctbW;
  const m355_ctb ctbinfo = p.ctbs[ctb];
  const int l2c = p.pp.log2_ctb_size;

  /* ---- wait for the neighbour CTBs whose INTRA output this CTB reads (left, above-left, above,
   * above-right).  The host derives the mask from the block lists (runtime.hip, intra_dependencies):
   * a neighbour matters only if it is in the same tile, one of our intra blocks touches the shared
   * border on our side, and one of its intra blocks touches it on its side.  Everything else next
   * door was finished by the preceding kernels (stream order), so sparse intra CUs in inter
   * pictures decode fully in parallel and only genuinely chained CTBs form a wavefront. ---- */
  const uint8_t dep = p.ctb_dep[ctb];
  /* 3x3 CTB neighbourhood facts, once per CTB: every availability test of intrapred.h:486-508 / :534-633
     (picture, slice, tile, z-scan order across CTBs) becomes an LDS lookup instead of dependent global loads */
  if (threadIdx.x >= 64 && threadIdx.x < 73) {
    const int i = threadIdx.x - 64, nx = ctbX + i % 3 - 1, ny = ctbY + i / 3 - 1;
    uint32_t ts = 0xFFFFFFFFu; uint8_t same = 0;
    if (nx >= 0 && ny >= 0 && nx < p.ctbW && ny < p.ctbH) {
      const int n = ny * p.ctbW + nx;
      ts = p.ctb_ts[n];
      same = p.slices[p.ctbs[n].slice_idx].slice_addr_rs == p.slices[ctbinfo.slice_idx].slice_addr_rs && p.tile_id[n] == p.tile_id[ctb];
    }
    s_nts[i] = ts; s_nsame[i] = same;
  }

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
  uint16_t* top = s_top[cs];
  uint16_t* body = s_body + (cs == 0 ? 0 : BODY_L + (cs - 1) * BODY_C);
  const int BODY_PITCH = cs == 0 ? BODY_PITCH_OF(MAXCTB) : BODY_PITCH_OF(CW_C);
  int16_t* resl = s_res + (cs == 0 ? 0 : RES_L + (cs - 1) * RES_C);
  const int RES_PITCH = cs == 0 ? MAXCTB : CW_C;
  uint16_t* raw = s_raw[wv];
  uint16_t* pp_ = s_p[wv];
  uint16_t* pf = s_f[wv];
  int* ref = s_ref[wv] + 32;

  if (comp) {
    /* Inter pictures (!DENSE: a handful of intra blocks per CTB, one wave per component): only the samples some block's
       border gathers — the row above it (x-1 .. x+2nT-1) and the column left of it (y .. y+2nT-1), intrapred.h:436-674 —
       are read at all, so only the vectors holding them are staged (a 64x64 CTB with two 8x8 intra blocks: ~6 of its 512
       luma vectors) instead of the whole CTB.  Blocks written later land in the same LDS tile as before. */
    if (!DENSE) {
      for (int y = lane; y < ch; y += 64) s_need[cs][y] = 0;
      wave_sync();
      const int nvr = cw >> 3;                                /* vectors per row */
      for (uint32_t k = lane; k < ctbinfo.ib_count; k += 64) {
        const uint32_t* r = (const uint32_t*)&p.ibs[ctbinfo.ib_start + k];
        const uint32_t w0 = r[0], w1 = r[1];
        if ((w1 & 0xFFu) != (uint32_t)c) continue;
        const int nT = 1 << ((w1 >> 8) & 0xFFu);
        const int lx = (int)(w0 & 0xFFFFu) - x0c, ly = (int)(w0 >> 16) - y0c;
        if (ly >= 1) {
          const int v0 = max(lx - 1, 0) >> 3, v1 = min((lx + 2 * nT - 1) >> 3, nvr - 1);
          if (v1 >= v0) atomicOr(&s_need[cs][ly - 1], ((2u << v1) - 1u) & ~((1u << v0) - 1u));
        }
        if (lx >= 1) {
          const uint32_t bit = 1u << ((lx - 1) >> 3);
          const int y1 = min(ly + 2 * nT, ch);
          for (int y = max(ly, 0); y < y1; y++) atomicOr(&s_need[cs][y], bit);
        }
      }
      wave_sync();
    }
    /* ---- stage the CTB and its halo in LDS: 8-sample vectors, all loads of a lane in flight at once ---- */
    {
      const int l2v = (l2c - csw) - 3;                       /* log2(vectors per row); cw >= 8 */
      const int nvec = ch << l2v;
      for (int idx = lane + 64 * g; idx < nvec; idx += 64 * G) {
        const int y = idx >> l2v, xv = (idx & ((1 << l2v) - 1)) * 8;
        if (!DENSE && !((s_need[cs][y] >> (xv >> 3)) & 1u)) continue;
        if (x0c + xv < pw && y0c + y < ph) {
          const PIX* src = plane + (size_t)(y0c + y) * stride + x0c + xv;
          uint4 v;
          if (sizeof(PIX) == 2) v = *(const uint4*)src;
          else {
            const uint2 b = *(const uint2*)src;
            v.x = (b.x & 0xFFu) | ((b.x & 0xFF00u) << 8); v.y = ((b.x >> 16) & 0xFFu) | ((b.x >> 8) & 0xFF0000u);
            v.z = (b.y & 0xFFu) | ((b.y & 0xFF00u) << 8); v.w = ((b.y >> 16) & 0xFFu) | ((b.y >> 8) & 0xFF0000u);
          }
          *(uint4*)(body + y * BODY_PITCH + BODY_X0 + xv) = v;
        }
      }
    }
  }

  /* ---- residual pre-pass: every wave fetches the residuals of the blocks it will process (same selection as the
     main loop below, so no other wave ever reads them) ---- */
  constexpr bool res_in_lds = DENSE;
  for (uint32_t kbase = 0; res_in_lds && kbase < ctbinfo.ib_count; kbase += 64) {
    uint32_t rw0 = 0, rw1 = 0xFF, rw2 = 0;
    int lv = -1;
    const int nvalid = min(64, (int)(ctbinfo.ib_count - kbase));
    if (lane < nvalid) {
      const uint32_t* r = (const uint32_t*)&p.ibs[ctbinfo.ib_start + kbase + lane];
      rw0 = r[0]; rw1 = r[1]; rw2 = r[2];
      lv = p.ib_level[ctbinfo.ib_start + kbase + lane];
    }
    const int lv_first = __shfl(lv, 0, 64), lv_last = __shfl(lv, nvalid - 1, 64);
    for (int L = lv_first; L <= lv_last; L++) {
      unsigned long long mine = __ballot((int)(comp && lv == L && (rw1 & 0xFFu) == (uint32_t)c));
      int rank = 0;
      while (mine) {
        const int src = __ffsll(mine) - 1;
        mine &= mine - 1;
        if ((rank++ % G) != g) continue;
        const uint32_t w0 = __shfl(rw0, src, 64), w1 = __shfl(rw1, src, 64), w2 = __shfl(rw2, src, 64);
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

  /* ---- only now wait for the neighbour CTBs: everything above (own samples, finished by the preceding
     kernels) overlapped with their work; what follows — the halo — is what they produce ---- */
  if (threadIdx.x == 0 && (dep & 15)) {
    const int dx[4] = {-1, -1, 0, 1}, dy[4] = {0, -1, -1, -1};
    for (int n = 0; n < 4; n++) {
      const int nx = ctbX + dx[n], ny = ctbY + dy[n];
      if (nx < 0 || ny < 0 || nx >= p.ctbW) continue;
      const int nb = ny * p.ctbW + nx;
      if (!((dep >> n) & 1)) continue;         /* that neighbour's intra output is never read here */
      unsigned spins = 0;
      while (__hip_atomic_load(&p.ctb_done[nb], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != p.epoch) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > SPIN_LIMIT) { atomicExch(p.timeout, 1u); break; }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();

  if (comp && g == 0) {
    if (x0c > 0)
      for (int y = lane; y < ch; y += 64)
        if (y0c + y < ph) body[y * BODY_PITCH + BODY_X0 - 1] = plane[(size_t)(y0c + y) * stride + x0c - 1];
    if (y0c > 0)
      for (int x = lane; x < 2 * cw + 1; x += 64) {
        const int xx = x0c - 1 + x;
        if (xx >= 0 && xx < pw) top[x] = plane[(size_t)(y0c - 1) * stride + xx];
      }
  }
  if (G > 1) __syncthreads(); else wave_sync();     /* CTB + halo staged by all waves of the component (G == 1: by this wave) */

#define SAMPLE(lx, ly) ((ly) < 0 ? top[(lx) + 1] : body[(ly) * BODY_PITCH + (lx) + BODY_X0])

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
    const int lv_first = __shfl(lv, 0, 64), lv_last = __shfl(lv, nvalid - 1, 64);
    for (int L = lv_first; L <= lv_last; L++) {
    unsigned long long mine = __ballot((int)(comp && lv == L && (rw1 & 0xFFu) == (uint32_t)c));
    int rank = 0;
    while (mine) {
      const int src = __ffsll(mine) - 1;
      mine &= mine - 1;
      if ((rank++ % G) != g) continue;             /* another wave of this component takes it */
      m355_ib ib;
      {
        const uint32_t w0 = __shfl(rw0, src, 64), w1 = __shfl(rw1, src, 64), w2 = __shfl(rw2, src, 64);
        ib.x = (uint16_t)(w0 & 0xFFFFu); ib.y = (uint16_t)(w0 >> 16);
        ib.cidx = (uint8_t)(w1 & 0xFFu); ib.log2_size = (uint8_t)((w1 >> 8) & 0xFFu); ib.mode = (uint8_t)((w1 >> 16) & 0xFFu); ib.flags = (uint8_t)(w1 >> 24);
        ib.res_ofs = w2;
      }
      const int nT = 1 << ib.log2_size;
      const int xB = ib.x, yB = ib.y, lx = xB - x0c, ly = yB - y0c;

      if (ib.flags & M355_IBF_PCM) { /* raw block */
        for (int o = lane; o < nT * nT; o += 64) {
          const int y = o >> ib.log2_size, x = o & (nT - 1);
          const uint16_t v = p.pcm[ib.res_ofs + o];
          body[(ly + y) * BODY_PITCH + lx + x + BODY_X0] = v;
          plane[(size_t)(yB + y) * stride + xB + x] = (PIX)v;
        }
        wave_sync();
        continue;
      }

      /* residual of this block (written by k_residual), sparse case: issue the loads now, consume them after the
         border/prediction chain — up to 16 samples per lane (32x32) */
      int16_t rv[16];
#pragma unroll
      for (int q = 0; q < 16; q++) {
        const int o = lane + 64 * q;
        rv[q] = (!res_in_lds && (ib.flags & M355_IBF_HAS_RESIDUAL) && o < nT * nT) ? p.resbuf[ib.res_ofs + o] : (int16_t)0;
      }

      /* ---- preproc (intrapred.h:436-531): CTB-level availability from the neighbourhood table ---- */
      const int xBL = xB * SubW, yBL = yB * SubH;
      bool aL = xBL != 0, aT = yBL != 0, aTL = xBL != 0 && yBL != 0, aTR = yBL != 0;
      if (xBL + nT * SubW >= p.pp.width) aTR = false;
      {
        const int dxL = ((xBL - 1) >> l2c) - ctbX, dxR = ((xBL + nT * SubW) >> l2c) - ctbX, dyT = ((yBL - 1) >> l2c) - ctbY;
        if (aL && !s_nsame[3 + dxL + 1]) aL = false;
        if (aT && !s_nsame[(dyT + 1) * 3 + 1]) aT = false;
        if (aTL && !s_nsame[(dyT + 1) * 3 + dxL + 1]) aTL = false;
        if (aTR && !s_nsame[(dyT + 1) * 3 + dxR + 1]) aTR = false;
      }
      int nBottom = p.pp.height - yB * SubH;
      nBottom = (nBottom + SubH - 1) / SubH;
      if (nBottom > 2 * nT) nBottom = 2 * nT;
      int nRight = p.pp.width - xB * SubW;
      nRight = (nRight + SubW - 1) / SubW;
      if (nRight > 2 * nT) nRight = 2 * nT;
      const int l2tb = p.pp.log2_min_tb_size, cmask = (1 << l2c) - 1;
      const uint32_t curTs = s_nts[4], curZ = d_morton((uint32_t)(xBL & cmask) >> l2tb, (uint32_t)(yBL & cmask) >> l2tb);
      const bool cip = (p.pp.flags & M355_PF_CONSTRAINED_INTRA_PRED) != 0;
      const int nEnt = 4 * nT + 1;

      /* ---- fill_from_image (intrapred.h:534-633): one border entry per lane and chunk ---- */
      unsigned long long am[3] = {0, 0, 0};
#pragma unroll
      for (int q = 0; q < 3; q++) {
        if (64 * q >= nEnt) continue;              /* wave-uniform: 4x4 / 8x8 blocks have 17 / 33 border entries */
        const int e = lane + 64 * q;
        bool av = false;
        int val = 0;
        if (e < nEnt) {
          const int i = e - 2 * nT;
          int xN, yN, sx, sy; /* test position (luma), sample position (local) */
          if (i < 0) {
            const int yy = -i - 1, g = yy & ~3;
            av = aL && (g + 3 < nBottom);
            xN = (xB - 1) * SubW; yN = (yB + g + 3) * SubH; sx = lx - 1; sy = ly + yy;
          } else if (i == 0) {
            av = aTL;
            xN = (xB - 1) * SubW; yN = (yB - 1) * SubH; sx = lx - 1; sy = ly - 1;
          } else {
            const int xx = i - 1, g = xx & ~3;
            av = (g < nT ? aT : aTR) && (g < nRight);
            xN = (xB + g) * SubW; yN = (yB - 1) * SubH; sx = lx + xx; sy = ly - 1;
          }
          if (av) {     /* MinTbAddrZS[neighbour] <= MinTbAddrZS[current] (intrapred.h:560-566) */
            const int dcx = (xN >> l2c) - ctbX, dcy = (yN >> l2c) - ctbY;
            if (dcx == 0 && dcy == 0) av = d_morton((uint32_t)(xN & cmask) >> l2tb, (uint32_t)(yN & cmask) >> l2tb) <= curZ;
            else av = s_nts[(dcy + 1) * 3 + dcx + 1] < curTs;
          }
          if (av && cip) av = d_is_intra_at(p, xN, yN);
          if (av) val = SAMPLE(sx, sy);
          raw[e] = (uint16_t)val;
        }
        am[q] = __ballot(av);
      }
      wave_sync();
      /* ---- reference_sample_substitution ---- */
      const bool none = (am[0] | am[1] | am[2]) == 0;
#pragma unroll
      for (int q = 0; q < 3; q++) {
        if (64 * q >= nEnt) continue;
        const int e = lane + 64 * q;
        if (e < nEnt) {
          int v;
          if (none) v = 1 << (bd - 1);
          else if ((am[q] >> lane) & 1) v = raw[e];
          else v = raw[d_subst_src(e, am[0], am[1], am[2])];
          pp_[e] = (uint16_t)v;
        }
      }
      wave_sync();
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
      const int angle = c_intra_angle[mode];
      if (mode >= 2) {
        const int sgn = mode >= 18 ? 1 : -1;
        const int inv = angle < 0 ? c_intra_inv_angle[mode - 11] : 0;
        const int lo = (nT * angle) >> 5;
        for (int t = lane; t < 3 * nT + 1; t += 64) {
          const int x = t - nT;
          int v = 0;
          if (x >= 0 && x <= nT) v = BRD(sgn * x);
          else if (x < 0) { if (angle < 0 && lo < -1 && x >= lo) v = BRD(-sgn * ((x * inv + 128) >> 8)); }
          else if (angle >= 0) v = BRD(sgn * x);
          ref[x] = v;
        }
        wave_sync();
      }
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
          v = iFact ? ((32 - iFact) * ref[b + iIdx + 1] + iFact * ref[b + iIdx + 2] + 16) >> 5 : ref[b + iIdx + 1];
          if (bfilt) {
            if (mode == 26 && x == 0) v = d_clip_bd(BRD(1) + ((BRD(-1 - y) - BRD(0)) >> 1), bd);
            if (mode == 10 && y == 0) v = d_clip_bd(BRD(-1) + ((BRD(1 + x) - BRD(0)) >> 1), bd);
          }
        }
        if (has_res) v = d_clip_bd(v + (res_in_lds ? (int)resl[(ly + y) * RES_PITCH + lx + x] : (int)rv[q]), bd);
        body[(ly + y) * BODY_PITCH + lx + x + BODY_X0] = (uint16_t)v;     /* for the next blocks' borders */
        plane[(size_t)(yB + y) * stride + xB + x] = (PIX)v;                /* the picture: only intra samples are (re)written */
      }
      wave_sync();
    }   /* this wave's blocks of the level */
    if (G > 1) __syncthreads();   /* level done: its samples are in LDS for the next level's borders (G == 1: the wave's own
                                     blocks are ordered by wave_sync above; components do not interact) */
    }   /* levels in the batch */
  }   /* 64-record batches */
#undef BRD
#undef SAMPLE

  /* ---- publish (guideline 16: stores -> barrier -> one-lane agent release -> drain -> flag) ---- */
  if (!(dep & 16)) return;            /* nobody waits for this CTB (host-derived): nothing to publish */
  d_drain_vmem();
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    d_drain_vmem();
    __hip_atomic_store(&p.ctb_done[ctb], p.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

template <class PIX, int CF>
static void launch_intra_cf(const DevPic& p, hipStream_t st)
{
  hipMemsetAsync(p.ticket, 0, 4, st);
  if (p.intra_waves >= INTRA_GMAX) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_intra<PIX, CF, true>), dim3(p.n_intra_work), dim3(192 * INTRA_GMAX), 0, st, p, 0, p.n_intra_work, 1);
  else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_intra<PIX, CF, false>), dim3(p.n_intra_work), dim3(192), 0, st, p, 0, p.n_intra_work, 1);
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
