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
 * (image.h:76-80, slice.cc:4789-4795) moved onto the device.  An intra picture is DEPENDENCY-bound: the 1080p
 * picture of BASELINE config 2 is a chain of ~1900 dependent block levels of which only ~60 cross a CTB
 * boundary, and a wave that runs alone on its SIMD retires one instruction every ~5 cycles — so what decides the
 * stage is the number of INSTRUCTIONS between two dependent blocks, not memory latency.  Hence two kernels:
 *
 *   k_intra_plan  (fully parallel, off the chain, runs beside k_inter / k_residual on the side stream): everything of
 *     a block that does not depend on sample VALUES — which of its 4nT+1 border entries are available (picture,
 *     slice, tile, z-scan order, constrained intra prediction: intrapred.h:436-633), where each entry's sample will
 *     live in the CTB's LDS tile, and the substitution of unavailable entries (intrapred.h:637-665) — is resolved into
 *     a PLAN: one 16-bit LDS source per border entry;
 *   k_intra  (the chain): one workgroup per CTB that contains intra blocks, described by one host-prepared 32-byte
 *     record (DevIntraWork).  The CTB's samples, its residuals (inside the same tile, at each block's own position, until the block
 *     is predicted) and its plan are resident in LDS; a block is: gather the
 *     border through the plan -> (smooth) -> predict -> add residual -> LDS.  No picture store, no metadata lookup and
 *     no availability arithmetic sits between two dependent blocks; the CTB's intra samples are written to the picture
 *     once, coalesced, when the CTB is finished.
 *   - inter pictures: the CTBs of dependency chains (they read a neighbour's intra samples, or a neighbour reads theirs) are claimed
 *     from an atomic ticket, longest remaining chain first — an order in which a producer precedes its readers, so a workgroup only ever
 *     waits on CTBs claimed before it —, the others are taken by workgroup index; intra pictures: every CTB through the ticket, those
 *     that wait for no neighbour first, then the dependent ones in wavefront order;
 *   - dependencies between CTBs are tracked at the granularity of the SAMPLES ACTUALLY READ: a block that finishes a
 *     piece of its CTB's right column or bottom row publishes those samples as 8-byte granules {tag = this decode's
 *     epoch, two samples} with agent-scope (write-through) stores — "the data is the flag" (MI355X guide, guideline
 *     16, form R2: no fences, no write-back of the L2).  A consumer stages its halo from the picture where the
 *     neighbouring samples come from the preceding kernels and from the granules where they come from an intra block
 *     of a neighbour CTB; a granule that is not there yet is polled only by the block whose border needs it;
 *   - inside the workgroup the blocks run LEVEL BY LEVEL (host-derived levels, runtime_upload.hip intra_schedule: blocks of
 *     one level are independent); each colour component has 1, 2, 4 or 8 wavefronts that share a level's blocks, with
 *     a workgroup barrier between levels when there is more than one wave per component;
 *   - the inverse transforms were done up front, in parallel, by k_residual.
 */
#include <stdlib.h>
#include <type_traits>
#include "k_common.h"
#include "k_meta_tu.h"

#include "k_intra_plan.h"
typedef unsigned long long m355_granule;   /* (epoch << 32) | sample1 << 16 | sample0 */

/* right-column granules of CTB column `col` of component c: index (row of the picture) >> 1 */
__device__ __forceinline__ m355_granule* d_edge_col(const DevPic& p, int c, int col, int y) { return p.edge + p.edge_col_ofs[c] + (size_t)col * (size_t)(p.ph[c] >> 1) + (size_t)(y >> 1); }
/* bottom-row granules of CTB row `row`: index (column of the picture) >> 1 */
__device__ __forceinline__ m355_granule* d_edge_row(const DevPic& p, int c, int row, int x) { return p.edge + p.edge_row_ofs[c] + (size_t)row * (size_t)(p.pw[c] >> 1) + (size_t)(x >> 1); }

/* A halo sample its CTB has not published yet: poll its granule (relaxed, agent scope: served by the L2 / fabric, never by this
 * CU's L1); every waiting lane has its own word, the wave leaves when all have arrived.  Returns the lane's (possibly updated)
 * value; arrived samples are also stored in the halo words for the blocks that read them later. */
__device__ __forceinline__ uint32_t d_poll_halo(const m355_granule* top_row, const m355_granule* left_col, uint32_t* timeout, uint16_t* halo, int hi, uint32_t val, bool pending, int x0c, int y0c, uint32_t epoch)
{
  const int lane = threadIdx.x & 63;
  const bool is_top = hi < HALO_TOP_N;
  const int pos = is_top ? x0c - 1 + hi : y0c + hi - HALO_TOP_N;      /* picture column of a top entry / row of a left entry */
  const m355_granule* gsrc = pending ? (is_top ? top_row : left_col) + (pos >> 1) : nullptr;   /* (pos >= 0 for a pending lane) */
  unsigned spins = 0;
  for (;;) {
    if (pending) {
      const m355_granule gr = __hip_atomic_load(gsrc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if ((uint32_t)(gr >> 32) == epoch) { val = (uint32_t)((gr >> (16 * (pos & 1))) & 0xFFFFu); halo[hi] = (uint16_t)val; pending = false; }
    }
    if (!__any(pending)) break;
    __builtin_amdgcn_s_sleep(1);
    if ((++spins & 1023u) == 0 && (spins > SPIN_LIMIT || __hip_atomic_load(timeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
      if (lane == 0) atomicExch(timeout, 1u);        /* a list that promises a sample nobody produces: ONE bounded wait per decode */
      break;
    }
  }
  return val;
}

template <int CF> __global__ void __launch_bounds__(256) k_intra_plan(DevPic p, int work_n) { k_intra_plan_body<CF>(p, work_n, (int)blockIdx.x, (int)blockIdx.y, (int)gridDim.y); }
template <int CF> __global__ void __launch_bounds__(256) k_intra_plan_batch(DevBatch b) { M355_BATCH_PIC(b); k_intra_plan_body<CF>(p, p.n_intra_work, (int)blockIdx.x, (int)blockIdx.y, (int)gridDim.y); }
/* EXPERIMENT (M355_MERGE_TU_PLAN=1, emulator-verified only): the transform-edge scatter (k_meta_tu.h) and the border plans as ONE launch —
   both only read what k_meta_planes wrote, and a launch costs about 2 us of pipeline time (profiles/r04_aj_*): blocks [0, nb_tu) walk the
   transform leaves, block nb_tu + item * n_parts + part plans part `part` of work item `item` */
template <int CF> __global__ void __launch_bounds__(256) k_tu_plan(DevPic p, int nb_tu, int work_n, int n_parts)
{
  const int b = (int)blockIdx.x;
  if (b < nb_tu) { k_meta_tu_body(p, b); return; }
  const int q = b - nb_tu;
  k_intra_plan_body<CF>(p, work_n, q / n_parts, q % n_parts, n_parts);
}

/* NW = waves per workgroup: 12 for intra pictures (CTBs with hundreds of blocks: up to 8 luma + 2 + 2 chroma waves share a
 * level, the CTB's whole plan is fetched into LDS up front; 13 with the halo keeper, one picture at a time), 4 for inter pictures
 * (a handful of intra blocks per CTB: up to 2 + 1 + 1 waves, the plan of 64 blocks at a time, 80 registers: six workgroups per CU).
 * Every instantiation runs the same block code — residuals in LDS inside the body tile, class-specialised 16x16 / 32x32 loops, a
 * 32x32 block shared by its component's waves — and differs in how CTBs are claimed (persistent workgroups and a ticket / one
 * workgroup per CTB, chain CTBs through the ticket) and in the prologue (an inter picture's: everything whose address follows from
 * the descriptor requested at once).  The CTB's own wave counts come from DevIntraWork.waves_code (runtime_upload.hip intra_schedule). */
/* BATCH (intra pictures only): SEVERAL pictures' CTBs in one launch — pics[0 .. n_pics) in device memory, one shared ticket; ticket t
 * is item t / n_pics of picture t % n_pics (the pictures' wavefronts interleaved: a workgroup still only waits on items claimed
 * before its own, now of its own picture).  Independent intra pictures then overlap CTB by CTB inside one kernel instead of through
 * the runtime's hardware queues (m355_decode_batch, runtime_decode.hip). */
#ifdef __clang__
#define M355_WAVES_PER_EU(n) __attribute__((amdgpu_waves_per_eu(n)))
#else
#define M355_WAVES_PER_EU(n)      /* (the CPU tier's interpreter build) */
#endif
/* (the 12-wave kernel of intra pictures in flight at 96 / 80 registers — 5 / 6 waves per SIMD, i.e. two workgroups per CU at 80 —: C2 with three in flight
   0.353 -> 0.351 / 0.388 ms, nine in flight 0.226 -> 0.224 / 0.210, batches of 8: 0.133 -> 0.131 / 0.136 (profiles/r05_v24_*): it stays at 4) */
#ifndef M355_INTRA_SPARSE_WAVES_PER_EU
#define M355_INTRA_SPARSE_WAVES_PER_EU 6   /* (the register budget of an inter picture's kernel at 4:2:0 / 4:0:0: 6 -> 80 registers and one spilled pair = 6 workgroups per CU, C5's intra stage 0.0467 -> 0.0424 ms, C3 / C4 unchanged: profiles/r05_v22_intra_w6_ab.txt; 4 -> 96 registers = 5 workgroups; 4:2:2 / 4:4:4 are bounded by their LDS) */
#endif
template <class PIX, int CF, int NW, bool DENSE, bool BATCH>
__global__ void __launch_bounds__(64 * NW) M355_WAVES_PER_EU((DENSE || CF >= 2) ? 4 : M355_INTRA_SPARSE_WAVES_PER_EU) k_intra(DevPic p0, int work_n, const DevPic* __restrict__ pics, int n_pics, uint32_t* batch_ticket)
{
  if (!BATCH) M355_GATE(p0);
  constexpr int CW_C = IntraGeo<CF>::CW_C;
  constexpr int BODY_L = IntraGeo<CF>::BODY_L, BODY_C = IntraGeo<CF>::BODY_C;
  constexpr int SAMP_L = IntraGeo<CF>::SAMP_L, SAMP_C = IntraGeo<CF>::SAMP_C;
  __shared__ __attribute__((aligned(16))) uint16_t s_body[SAMP_L + 2 * SAMP_C];   /* per component: body | halo | constant cell */
  /* the plan (k_intra_plan): the whole CTB's (dense: M355_INTRA_PLAN_CAP entries, runtime.hip rejects CTBs beyond it) or 64
     blocks' at a time (M355_INTRA_PLAN_BATCH) */
  constexpr int PLAN_LDS = DENSE ? M355_INTRA_PLAN_CAP(CF) : M355_INTRA_PLAN_BATCH(CF);
  /* ONE intra picture at a time (the instantiation with the halo keeper, DevPic.intra_keeper): its CTBs are planned by k_intra itself
     (below).  With pictures in flight the planner's launch stays: it plans at the whole GPU's rate beside the other pictures' kernels,
     whereas planning inside the persistent workgroups adds to what bounds them then (C2, planned here: one at a time 0.920 -> 0.881 ms,
     three in flight 0.358 -> 0.378, batches of 8 0.131 -> 0.177: profiles/r05_v20_*) */
  constexpr bool PLAN_HERE = DENSE && NW == M355_INTRA_KEEPER_NW;
  __shared__ __attribute__((aligned(16))) uint16_t s_plan[PLAN_LDS];
  /* per WAVE (a wave works on one block at a time): */
  __shared__ uint16_t s_raw[NW][4 * 32 + 8];      /* gathered border, entry e = i + 2nT */
  __shared__ uint16_t s_f[NW][4 * 32 + 8];        /* filtered border */
  __shared__ uint32_t s_ticket;
  __shared__ uint32_t s_need[3][MAXCTB];         /* per component and CTB row, which 8-sample vectors some block's border reads */
  __shared__ uint32_t s_cover[3][MAXCTB / 4];    /* per component and row of 4x4 units, which units an intra block of this CTB writes */
  __shared__ uint32_t s_hneed[3][8];             /* ... and which halo entries (bit h: top entries 0 .. 2cw, then the left column) */
  /* the halo keeper's slots (below): granule address and halo element per lane and slot */
  __shared__ uint32_t s_kp_idx[(DENSE && NW == M355_INTRA_KEEPER_NW) ? 5 * 64 : 1];      /* (an index into DevPic.edge: the load stays a GLOBAL one — a pointer out of LDS makes it flat, and a flat load also counts as an LDS operation, which the wave waits for in front of every barrier) */
  __shared__ uint16_t s_kp_h1[(DENSE && NW == M355_INTRA_KEEPER_NW) ? 5 * 64 : 1];

  /* everything derived from the wave index or from a block record is wave-uniform: say so (readfirstlane / readlane), so that
     the component's plane pointers, pitches and granule offsets are scalar loads from the kernel arguments instead of vector
     loads, record fields live in SGPRs and the per-block branches are scalar */
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;

  /* Work items.  Intra pictures (DENSE): the workgroup is PERSISTENT — it claims CTB after CTB through the ticket, in work-list
     order (CTBs that wait for no neighbour first, then the dependent ones in wavefront order), until the list is exhausted.  The
     grid is therefore free to be smaller than the list (runtime.hip: with several pictures in flight each picture gets a share of
     the GPU's workgroup slots that covers its ACTIVE wavefront, instead of parking a workgroup on every CTB of the picture for
     the picture's whole duration); a workgroup only ever waits on items claimed before its own, whose workgroups are running.
     Inter pictures: one workgroup per CTB with intra blocks — the first n_intra_ticket items (the CTBs of dependency chains, the
     longest remaining chain first) through the ticket, taken by the workgroups of lowest index, i.e. the first to be dispatched; the
     CTBs without dependencies by workgroup index (no atomic, no barrier). */
  for (;;) {
  int item = (int)blockIdx.x, pk = 0;
  if (BATCH) {
    if (threadIdx.x == 0) s_ticket = atomicAdd(batch_ticket, 1u);
    __syncthreads();
    const uint32_t t = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_ticket);
    if (t >= (uint32_t)work_n * (uint32_t)n_pics) return;       /* work_n = the longest picture's list */
    pk = (int)(t % (uint32_t)n_pics); item = (int)(t / (uint32_t)n_pics);
  } else if (DENSE || item < p0.n_intra_ticket) {         /* (uniform per workgroup) */
    if (threadIdx.x == 0) s_ticket = atomicAdd(p0.ticket, 1u);
    __syncthreads();
    item = __builtin_amdgcn_readfirstlane((int)s_ticket);
  }
  item = __builtin_amdgcn_readfirstlane(item);              /* (uniform either way: say so — the descriptor's address is then a scalar one) */
  const DevPic& p = *(BATCH ? pics + pk : &p0);
  if (BATCH) {
    /* a shorter picture's list is exhausted, or its lists were rejected (k_validate): nothing to do for this ticket */
    if (item >= p.n_intra_work || p.timeout[1] == p.epoch) { __syncthreads(); continue; }
  } else if (item >= work_n) return;
  /* the CTB's descriptor (host-prepared, runtime.hip): one scalar 32-byte load */
  const DevIntraWork* wp = p.intra_work + item;
  const uint4 wd0 = *(const uint4*)wp;
  const uint4 wd1 = ((const uint4*)wp)[1];
  const int ctb = __builtin_amdgcn_readfirstlane((int)wd0.x);
  struct { uint32_t ib_start, ib_count; } ctbinfo = {(uint32_t)__builtin_amdgcn_readfirstlane((int)wd0.y), (uint32_t)__builtin_amdgcn_readfirstlane((int)wd0.z)};
  const uint32_t plan_base = (uint32_t)__builtin_amdgcn_readfirstlane((int)wd1.y), plan_count = (uint32_t)__builtin_amdgcn_readfirstlane((int)wd1.z);
  /* wave -> (colour component c, sub-wave g of G): GL luma waves, then GC for Cb, GC for Cr; the rest have no part in this CTB:
     a one-CTB workgroup's leave at once (finished waves do not take part in later barriers), a persistent workgroup's run along
     as waves of an absent component (they execute the barriers and take the next CTB with the others) */
  int GL, GC;
  {
    const int code = __builtin_amdgcn_readfirstlane((int)(wd1.x & 3u));   /* widest level of the CTB: luma 1 / 2 / 3-4 / more blocks */
    constexpr int GLMAX = NW >= 12 ? 8 : (NW >= 6 ? 4 : 2), GCMAX = (NW - GLMAX) / 2 >= 4 ? 4 : ((NW - GLMAX) / 2 >= 2 ? 2 : 1);
    GL = min(GLMAX, code == 0 ? 1 : (code == 1 ? 2 : (code == 2 ? 4 : 8)));
    GC = code == 3 ? GCMAX : 1;
    /* intra pictures: never fewer than two waves per component — a component's blocks go round its waves (below), so that
       a wave prepares its next block while another one predicts the current level's */
    if (DENSE) { GL = max(GL, 2); GC = GCMAX; }
  }
  const bool spare = wv >= GL + 2 * GC;
  if (spare && !DENSE) return;
  const int c = spare ? 3 : (wv < GL ? 0 : (wv < GL + GC ? 1 : 2));
  const int G = spare ? 1 : (c == 0 ? GL : GC), g = spare ? 0 : (c == 0 ? wv : (wv - GL - (c - 1) * GC));
  const int NWV = DENSE ? NW : GL + 2 * GC;              /* waves running this CTB's loops */
  const bool multi = GL + GC > 2;                        /* more than one wave per component somewhere: levels end in a barrier */
  const int ctbX = ctb % p.ctbW, ctbY = ctb / p.ctbW;
  const int l2c = p.pp.log2_ctb_size;

  const int nc = p.pp.chroma_format_idc ? 3 : 1;
  const bool comp = c < nc;
  const int csw = (c && comp) ? (p.sw == 2) : 0, csh = (c && comp) ? (p.sh == 2) : 0;
  const int cw = (1 << l2c) >> csw, ch = (1 << l2c) >> csh;
  const int x0c = (ctbX << l2c) >> csw, y0c = (ctbY << l2c) >> csh;
  const int bd = (c && comp) ? p.pp.bit_depth_chroma : p.pp.bit_depth_luma;
  const int cs = comp ? c : 0;
  PIX* plane = (PIX*)p.plane[cs];
  const int stride = p.stride[cs], pw = p.pw[cs], ph = p.ph[cs];
  uint16_t* body = s_body + (cs == 0 ? 0 : SAMP_L + (cs - 1) * SAMP_C);
  const int HALO_BASE = cs == 0 ? BODY_L : BODY_C;
  uint16_t* halo = body + HALO_BASE;
  const int BODY_PITCH = cs == 0 ? BODY_PITCH_OF(MAXCTB) : BODY_PITCH_OF(CW_C);
  uint16_t* raw = s_raw[wv];
  uint16_t* pf = s_f[wv];
  const uint32_t epoch = p.epoch;
#define SYNC_CTB() do { if (multi) __syncthreads(); else wave_sync(); } while (0)

  /* ---- The prologue is a handful of DEPENDENT memory round trips (descriptor -> block records -> CU plane -> CU records -> samples), and on an
     inter picture it is half of a CTB's time in the kernel (profiles/r05_g_intra_sparse_timeline.txt: 6 of 12 us).  EARLY = everything whose address
     follows from the descriptor alone is requested at once, before the first wait: the first 64 block records and exec records (an inter picture's
     CTB rarely has more) and the CU index behind every halo entry; the CU records follow beside the plan, the residuals beside the need scan, the
     halo samples / granules beside the body.  (An intra picture's prologue is off the chain — a CTB is claimed long before its neighbours let it
     run — and its kernels have no registers to spare across the in-kernel planning: there everything is loaded where it is used.) ---- */
  constexpr bool EARLY = !DENSE;
  constexpr int HU = 4;                                    /* halo entries per lane: 2cw + 1 + ch <= 193 entries on >= 64 lanes */
  static_assert(3 * MAXCTB + 1 <= 64 * HU, "a component's halo is staged in one pass");
  const int nhalo = (2 * cw + 1) + ch;
  uint4 ex_first = make_uint4(0, 0, 0, 0);
  uint32_t rf0 = 0, rf1 = 0xFFu, rf2 = 0;
  if (EARLY && lane < (int)min(ctbinfo.ib_count, 64u)) {
    ex_first = ((const uint4*)p.ib_aux)[ctbinfo.ib_start + lane];
    const uint32_t* r = (const uint32_t*)&p.ibs[ctbinfo.ib_start + lane];
    rf0 = r[0]; rf1 = r[1]; rf2 = r[2];
  }
  /* halo entry h of this lane's u-th slot: picture position, inside the picture?, CU index / produced by an intra block? */
  int hx[HU], hy[HU];
  bool hin[HU], htop[HU], hintra[HU];
  uint32_t hci[HU], hpm[HU];
  auto halo_where = [&]() {
#pragma unroll
    for (int u = 0; u < HU; u++) {
      const int h = lane + 64 * g + u * 64 * G;
      htop[u] = h < 2 * cw + 1;
      hx[u] = htop[u] ? x0c - 1 + h : x0c - 1; hy[u] = htop[u] ? y0c - 1 : y0c + (h - (2 * cw + 1));
      hin[u] = comp && h < nhalo && hx[u] >= 0 && hy[u] >= 0 && hx[u] < pw && hy[u] < ph;
      /* (loads without a branch around them — an entry outside the picture reads element 0 —: the four are in flight together, where a
         conditional load is waited for inside its branch) */
      hci[u] = d_cu_index_at(p, hin[u] ? hx[u] << csw : 0, hin[u] ? hy[u] << csh : 0);
    }
  };
  auto halo_who = [&]() {
#pragma unroll
    for (int u = 0; u < HU; u++) hpm[u] = p.cus[hci[u] ? hci[u] - 1 : 0u].pred_mode;
  };
  auto halo_is_intra = [&]() {
#pragma unroll
    for (int u = 0; u < HU; u++) hintra[u] = hin[u] && hci[u] != 0 && hpm[u] == 0;
  };
  if (EARLY) halo_where();

  /* ---- the plan: the whole CTB's (dense) or its first PLAN_LDS entries (a later batch of 64 blocks reloads), 16 bytes per
     lane and step ---- */
  uint32_t plan_lo = 0;                                    /* first entry held in s_plan */
  if (PLAN_HERE) {
    /* the CTB is planned HERE, straight into LDS, by all the workgroup's waves (a block each, round robin): the prologue is off the
       chain — a CTB is claimed long before its neighbours let it run —, whereas the planner's launch stands in front of k_intra with
       45-78 us for the 66 000 blocks of the 1080p picture of config 2 (profiles/r05_z_c2_*) */
    const uint32_t nbs = (uint32_t)__builtin_amdgcn_readfirstlane((int)(wd0.w & 0xFFFFu)), nbe = (uint32_t)__builtin_amdgcn_readfirstlane((int)(wd0.w >> 16));
    for (uint32_t k = (uint32_t)wv; k < ctbinfo.ib_count; k += (uint32_t)NW)
      d_intra_plan_block<CF>(p, ctbinfo.ib_start + k, ctbX, ctbY, nbs, nbe, s_raw[wv], s_plan);
  } else {
    const uint32_t n = min(plan_count, (uint32_t)PLAN_LDS);
    const uint4* src = (const uint4*)(p.iplan + plan_base);  /* plan_base is a multiple of 8 entries */
    for (uint32_t o = (uint32_t)(wv * 64 + lane) * 8u; o < n; o += (uint32_t)NWV * 512u) *(uint4*)(s_plan + o) = src[o >> 3];
  }
  if (EARLY) halo_who();                                   /* (the CU indices came in with the plan) */

  /* ---- residual pre-pass: the CTB's deferred residuals go to LDS, each component's waves taking its blocks in turn (inter
     pictures too: a block of the chain then reads its residual from LDS, where it paid a global-memory round trip — and the
     32x32 blocks that make the chains of such a picture are predicted by the same loops as an intra picture's).  They go INTO
     THE BODY, at the block's own position: nothing reads those elements before the block is predicted — the CTB's blocks are
     disjoint, the staging below leaves covered units alone, a border entry only ever points at a sample that has been
     reconstructed — and the lane that predicts a sample reads its residual from the element it then overwrites.  Two blocks at a
     time (4 samples = one row segment per lane and step, up to four steps per block): their loads are in flight together. ---- */
  {
    int taken = 0;                                           /* blocks of this component seen so far (wave-uniform) */
    for (uint32_t kbase = 0; kbase < ctbinfo.ib_count; kbase += 64) {
      uint32_t rw0 = rf0, rw1 = rf1, rw2 = rf2;
      if (!(EARLY && kbase == 0)) {
        rw0 = 0; rw1 = 0xFFu; rw2 = 0;
        if (kbase + lane < ctbinfo.ib_count) {
          const uint32_t* r = (const uint32_t*)&p.ibs[ctbinfo.ib_start + kbase + lane];
          rw0 = r[0]; rw1 = r[1]; rw2 = r[2];
        }
      }
      unsigned long long mine = __ballot((int)(comp && (rw1 & 0xFFu) == (uint32_t)c));
      /* this wave's next block with a residual among the batch's records: first body element, size, residual offset (wave-uniform) */
      auto next_block = [&](int& b_ofs, int& b_log2, uint32_t& b_res) -> bool {
        while (mine) {
          const int src = __ffsll(mine) - 1;
          mine &= mine - 1;
          if ((taken++ & (G - 1)) != g) continue;
          const uint32_t w0 = __builtin_amdgcn_readlane(rw0, src), w1 = __builtin_amdgcn_readlane(rw1, src);
          const int flags = (int)(w1 >> 24);
          if (!(flags & M355_IBF_HAS_RESIDUAL) || (flags & M355_IBF_PCM)) continue;
          b_log2 = (int)((w1 >> 8) & 0xFFu); b_res = __builtin_amdgcn_readlane(rw2, src);
          b_ofs = ((int)(w0 >> 16) - y0c) * BODY_PITCH + BODY_X0 + (int)(w0 & 0xFFFFu) - x0c;
          return true;
        }
        return false;
      };
      auto load_block = [&](uint2* v, int b_log2, uint32_t b_res) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int o = lane * 4 + 256 * u;
          v[u] = make_uint2(0, 0);
          if (o < (1 << (2 * b_log2))) v[u] = *(const uint2*)(p.resbuf + b_res + o);
        }
      };
      auto store_block = [&](const uint2* v, int b_ofs, int b_log2) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int o = lane * 4 + 256 * u;
          if (o < (1 << (2 * b_log2))) *(uint2*)(body + b_ofs + (o >> b_log2) * BODY_PITCH + (o & ((1 << b_log2) - 1))) = v[u];
        }
      };
      for (;;) {
        int ofs_a = 0, log2_a = 2, ofs_b = 0, log2_b = 2;
        uint32_t res_a = 0, res_b = 0;
        if (!next_block(ofs_a, log2_a, res_a)) break;
        const bool two = next_block(ofs_b, log2_b, res_b);
        uint2 va[4], vb[4];
        load_block(va, log2_a, res_a);
        if (two) load_block(vb, log2_b, res_b);
        store_block(va, ofs_a, log2_a);
        if (two) store_block(vb, ofs_b, log2_b);
      }
    }
  }
  /* ---- which body vectors does some block's border read?  Only those are staged: the row above a block
     (x-1 .. x+2nT-1) and the column left of it (y .. y+2nT-1), intrapred.h:436-674 — a 64x64 CTB with two 8x8 intra
     blocks: ~6 of its 512 luma vectors — and of those only what no intra block of this CTB produces itself (s_cover:
     those samples reach the same LDS tile when their block is predicted; they are also what the CTB writes to the
     picture at the end). ---- */
  if (comp && g == 0) {
    for (int y = lane; y < ch; y += 64) s_need[cs][y] = 0;
    if (lane < 8) s_hneed[cs][lane] = 0;
    if (lane == 0) halo[HALO_N] = (uint16_t)(1u << (bd - 1));   /* the constant cell */
    if (lane < MAXCTB / 4) s_cover[cs][lane] = 0;
    wave_sync();
    const int nvr = cw >> 3;                                /* vectors per row */
    for (uint32_t kbase = 0; kbase < ctbinfo.ib_count; kbase += 64) {
      uint32_t w0 = rf0, w1 = rf1;
      if (!(EARLY && kbase == 0)) {
        w1 = 0xFFu;
        if (kbase + lane < ctbinfo.ib_count) {
          const uint32_t* r = (const uint32_t*)&p.ibs[ctbinfo.ib_start + kbase + lane];
          w0 = r[0]; w1 = r[1];
        }
      }
      if ((w1 & 0xFFu) != (uint32_t)c) continue;             /* (another component's block, or no block: 0xFF) */
      const int nT = 1 << ((w1 >> 8) & 0xFFu);
      const int lx = (int)(w0 & 0xFFFFu) - x0c, ly = (int)(w0 >> 16) - y0c;
      {
        const uint32_t um = ((1u << (nT >> 2)) - 1u) << (lx >> 2);
        for (int uy = ly >> 2; uy < (ly + nT) >> 2; uy++) atomicOr(&s_cover[cs][uy], um);
      }
      if ((w1 >> 24) & M355_IBF_PCM) continue;               /* raw blocks read no border */
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
    /* ---- halo: the row above the CTB (x = -1 .. 2cw-1) and the column left of it, only the entries some block's border reads
       (s_hneed).  A sample that an INTRA block of a neighbour CTB produces comes from that CTB's granules — if it is there already;
       otherwise the entry stays HALO_NOT_READY and the block that needs it polls for it.  Everything else was finished by the preceding
       kernels (stream order) and is read from the picture.  Requested here, stored behind the body's first pass. ---- */
    if (!EARLY) { halo_where(); halo_who(); }
    halo_is_intra();
    m355_granule hgr[HU];
    uint32_t hpl[HU];
    /* (scalars: the granule rows of the CTB row above / the CTB column on the left — a per-lane choice between DevPic's arrays would be a vector
       load from the kernel arguments in front of every granule load) */
    const size_t g_row = p.edge_row_ofs[cs] + (size_t)max(ctbY - 1, 0) * (size_t)(pw >> 1), g_col = p.edge_col_ofs[cs] + (size_t)max(ctbX - 1, 0) * (size_t)(ph >> 1);
#pragma unroll
    for (int u = 0; u < HU; u++) {
      const int h = lane + 64 * g + u * 64 * G;
      const bool want = hin[u] && ((s_hneed[cs][min(h, nhalo - 1) >> 5] >> (h & 31)) & 1u);
      hintra[u] = hintra[u] && want;
      /* (both loads of every slot, without a branch around them — a slot that wants neither reads element 0 —: eight loads in flight, where a
         conditional load is waited for inside its branch) */
      const size_t gi = htop[u] ? g_row + (size_t)(hx[u] >> 1) : g_col + (size_t)(hy[u] >> 1);     /* (d_edge_row / d_edge_col) */
      hgr[u] = __hip_atomic_load(p.edge + (hintra[u] ? gi : (size_t)0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      hpl[u] = plane[(want && !hintra[u]) ? (size_t)hy[u] * stride + hx[u] : (size_t)0];
      if (!want || hintra[u]) hpl[u] = 0;
    }
    /* ---- stage the needed body vectors: 8 samples each; four per lane are requested before the first is stored (a loop of
       load -> store steps costs one memory round trip per step, and a CTB has up to eight steps per lane) ---- */
    {
      const int l2v = (l2c - csw) - 3;                       /* log2(vectors per row); cw >= 8 */
      const int nvec = ch << l2v;
      constexpr int U = 4;
      /* (the first pass stands outside the loop: at a loop header the compiler waits for every load in flight — the halo's —, and an inter
         picture's chroma components have one pass) */
      auto stage_pass = [&](const int idx0) {
        uint4 v[U];
        bool take[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
          const int idx = idx0 + u * 64 * G;
          const int y = idx >> l2v, xv = (idx & ((1 << l2v) - 1)) * 8;
          const int yc = min(y, ch - 1);
          take[u] = idx < nvec && ((s_need[cs][yc] >> (xv >> 3)) & 1u) && ((s_cover[cs][yc >> 2] >> (xv >> 2)) & 3u) != 3u && x0c + xv < pw && y0c + y < ph;
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
          /* (a half some intra block of this CTB covers holds that block's RESIDUAL — above —, not a sample of the picture) */
          const uint32_t cov = (s_cover[cs][y >> 2] >> (xv >> 2)) & 3u;
          uint16_t* dst = body + y * BODY_PITCH + BODY_X0 + xv;
          if (cov == 0u) *(uint4*)dst = o;
          else if (cov == 2u) *(uint2*)dst = make_uint2(o.x, o.y);
          else *(uint2*)(dst + 4) = make_uint2(o.z, o.w);
        }
      };
      stage_pass(lane + 64 * g);
      for (int idx0 = lane + 64 * g + 64 * G * U; idx0 < nvec; idx0 += 64 * G * U) stage_pass(idx0);
    }
#pragma unroll
    for (int u = 0; u < HU; u++) {
      const int h = lane + 64 * g + u * 64 * G;
      if (h >= nhalo) continue;
      uint32_t val = hpl[u];
      if (hintra[u]) val = ((uint32_t)(hgr[u] >> 32) == epoch && !p.test_halo_late) ? (uint32_t)((hgr[u] >> (16 * ((htop[u] ? hx[u] : hy[u]) & 1))) & 0xFFFFu) : HALO_NOT_READY;
      if (htop[u]) halo[h] = (uint16_t)val; else halo[HALO_TOP_N + h - (2 * cw + 1)] = (uint16_t)val;
    }
  }
  __syncthreads();     /* bodies, halos, residuals and the plan staged */
  /* ---- the HALO KEEPER (intra pictures, the workgroup's last wave): a CTB's halo is staged long before its neighbours have
     finished (the prologue is off the chain), so nearly every sample a block reads from another CTB is still HALO_NOT_READY there,
     and the block's own poll costs it one fabric round trip even when the granule arrived long ago — on half of the levels of an
     intra picture (tools/intra_sim.py).  This wave holds the granules still missing (one per lane and slot), and once per level —
     beside the blocks, in front of the same barrier — takes in the loads it issued a level earlier, stores the samples that have
     arrived into the halo and asks again for the rest.  Its loads stay in flight across the barrier (raw s_barrier, no fence: gfx950
     backs off a barrier with memory operations outstanding).  A block whose sample is still missing polls as before. ---- */
  constexpr bool KEEPER = DENSE && NW == M355_INTRA_KEEPER_NW;
  constexpr int KSLOTS = CF == 3 ? 5 : 4;                    /* granules per lane: (cw + 1) + ch / 2 per component */
  const bool keeper = KEEPER && wv == NW - 1;
  uint32_t kp_pend = 0, kp_fly = 0;                          /* bit k: slot k is missing / has a load in flight */
  m355_granule kp_gr[KSLOTS];
  if (keeper) {
    const int cwl = 1 << l2c;
#pragma unroll
    for (int k = 0; k < KSLOTS; k++) kp_gr[k] = 0;
#pragma unroll 1
    for (int k = 0; k < KSLOTS; k++) {
      int q = lane + 64 * k, cq = -1, cwq = 0;
      for (int cc = 0; cc < nc; cc++) {
        const int cw_ = cc ? cwl >> (p.sw == 2) : cwl, ch_ = cc ? cwl >> (p.sh == 2) : cwl, n_ = cw_ + 1 + (ch_ >> 1);
        if (cq < 0) { if (q < n_) { cq = cc; cwq = cw_; } else q -= n_; }
      }
      if (cq < 0) continue;
      const int cswq = cq ? (p.sw == 2) : 0, cshq = cq ? (p.sh == 2) : 0;
      const int x0q = (ctbX << l2c) >> cswq, y0q = (ctbY << l2c) >> cshq;
      const int hbase = (cq == 0 ? 0 : SAMP_L + (cq - 1) * SAMP_C) + (cq == 0 ? BODY_L : BODY_C);
      const bool top = q <= cwq;
      const int t = top ? q : q - (cwq + 1);
      /* top granule t: picture columns x0 - 2 + 2t, + 1 = halo entries 2t - 1, 2t; left granule t: rows y0 + 2t, + 1 */
      const int h1 = top ? 2 * t : HALO_TOP_N + 2 * t + 1;
      const bool lo = !(top && t == 0);
      const bool miss = s_body[hbase + h1] == HALO_NOT_READY || (lo && s_body[hbase + h1 - 1] == HALO_NOT_READY);
      if (!miss) continue;                                   /* (a missing entry lies inside the picture and has a CTB above / left of it) */
      const size_t eofs = top ? (cq == 0 ? p.edge_row_ofs[0] : (cq == 1 ? p.edge_row_ofs[1] : p.edge_row_ofs[2]))
                              : (cq == 0 ? p.edge_col_ofs[0] : (cq == 1 ? p.edge_col_ofs[1] : p.edge_col_ofs[2]));
      const int pwq = cq == 0 ? p.pw[0] : (cq == 1 ? p.pw[1] : p.pw[2]), phq = cq == 0 ? p.ph[0] : (cq == 1 ? p.ph[1] : p.ph[2]);
      s_kp_idx[k * 64 + lane] = (uint32_t)(top ? eofs + (size_t)(ctbY - 1) * (size_t)(pwq >> 1) + (size_t)((x0q - 2 + 2 * t) >> 1)
                                               : eofs + (size_t)(ctbX - 1) * (size_t)(phq >> 1) + (size_t)((y0q + 2 * t) >> 1));
      s_kp_h1[k * 64 + lane] = (uint16_t)((hbase + h1) | (lo ? 0x8000 : 0));      /* (a component's arrays end below element 32768) */
      kp_pend |= 1u << k;
    }
  }
  auto keeper_step = [&]() {
    if (!__any((int)(kp_pend != 0u))) return;
    uint32_t gi[KSLOTS], hh[KSLOTS];
#pragma unroll
    for (int k = 0; k < KSLOTS; k++) { gi[k] = s_kp_idx[k * 64 + lane]; hh[k] = s_kp_h1[k * 64 + lane]; }   /* (one LDS round trip for all slots) */
#pragma unroll
    for (int k = 0; k < KSLOTS; k++) {
      if (!((kp_fly >> k) & 1u)) continue;
      const m355_granule gr = kp_gr[k];
      if ((uint32_t)(gr >> 32) != epoch) continue;
      s_body[hh[k] & 0x7FFFu] = (uint16_t)(gr >> 16);
      if (hh[k] & 0x8000u) s_body[(hh[k] & 0x7FFFu) - 1] = (uint16_t)gr;
      kp_pend &= ~(1u << k);
    }
    kp_fly = kp_pend;
#pragma unroll
    for (int k = 0; k < KSLOTS; k++)
      if ((kp_pend >> k) & 1u) kp_gr[k] = __hip_atomic_load(p.edge + gi[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  /* The CTB's exec records (runtime_upload.hip intra_schedule: sorted by level, then component; everything about a block that is not a
     sample value) are fetched 64 at a time (one per lane, 16 bytes, coalesced) by EVERY wave.  A wave's blocks of the batch —
     those of its component whose rank inside their (level, component) group falls to it — are one 64-bit mask; it walks them
     level by level, and all waves meet at the workgroup barrier after every level of the batch (the loop bounds come from the
     records alone, so the waves of absent components execute the same barriers).  What stands between two dependent blocks is
     kept short: the plan entries of the NEXT block's border (LDS reads that depend on no sample) are requested while the current
     block is predicted, a 4x4 / 8x8 block's border, smoothing and taps live in registers (cross-lane reads), its arithmetic is
     specialised by size (compile time) and mode class (one scalar branch). */
  const int thr_strong = 1 << (p.pp.bit_depth_luma - 5);
  const int pix_max = (1 << bd) - 1;
  /* a lane's sample of a 4x4 / 8x8 block, relative to the block's first sample in the body / residual tile */
  const int lofs_b4 = ((lane & 15) >> 2) * BODY_PITCH + (lane & 3), lofs_b8 = (lane >> 3) * BODY_PITCH + (lane & 7);   /* (4x4: the lanes beyond the block alias its samples — they read, never store) */
  int taken = 0;                                           /* blocks of this wave's component in earlier batches */
  if (KEEPER && keeper) {
    /* the halo keeper's walk through the same barriers (a path of its own: what it holds is live nowhere in the block code) */
    for (uint32_t kbase = 0; kbase < ctbinfo.ib_count; kbase += 64) {
      const int nvalid = min(64, (int)(ctbinfo.ib_count - kbase));
      const int lv = lane < nvalid ? (int)((p.ib_aux[4 * (size_t)(ctbinfo.ib_start + kbase + lane) + 3] >> 16) & 0x3FFFu) : -1;
      const int lv_first = __builtin_amdgcn_readlane(lv, 0), lv_last = __builtin_amdgcn_readlane(lv, nvalid - 1);
      for (int L = lv_first; L <= lv_last; L++) {
        keeper_step();          /* (every level: asking again only every 1.5 us was 3 % slower, profiles/r05_v15_*) */
        d_drain_lds(); __builtin_amdgcn_s_barrier();
      }
    }
  } else
  for (uint32_t kbase = 0; kbase < ctbinfo.ib_count; kbase += 64) {
    uint4 ex = make_uint4(0, 0, 0, 0);
    int lv = -1;
    const int nvalid = min(64, (int)(ctbinfo.ib_count - kbase));
    if (lane < nvalid) {
      ex = (EARLY && kbase == 0) ? ex_first : ((const uint4*)p.ib_aux)[ctbinfo.ib_start + kbase + lane];
      lv = (int)((ex.w >> 16) & 0x3FFFu);
    }
    if (!DENSE) {
      /* this batch's plan entries: [first block's offset, + PLAN_LDS) — reload when the window in LDS does not hold them */
      const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)ex.w, 0) & 0xFFFFu;
      const uint32_t last = (uint32_t)__builtin_amdgcn_readlane((int)ex.w, nvalid - 1) & 0xFFFFu;
      if (last + 130u > plan_lo + (uint32_t)PLAN_LDS && plan_count > (uint32_t)PLAN_LDS) {
        __syncthreads();                                     /* everybody is done with the previous window */
        const uint32_t lo8 = lo & ~7u;
        const uint32_t n = min(plan_count - lo8, (uint32_t)PLAN_LDS);
        const uint4* src = (const uint4*)(p.iplan + plan_base + lo8);
        for (uint32_t o = (uint32_t)(wv * 64 + lane) * 8u; o < n; o += (uint32_t)NWV * 512u) *(uint4*)(s_plan + o) = src[o >> 3];
        plan_lo = lo8;
        __syncthreads();
      }
    }
    const int lv_first = __builtin_amdgcn_readlane(lv, 0), lv_last = __builtin_amdgcn_readlane(lv, nvalid - 1);
    unsigned long long mine;
    {
      /* a component's blocks go ROUND ITS WAVES in record order (level by level): consecutive levels of a dependency chain fall
         to different waves, so the wave that predicts level L is not the one that has to fetch and decode the record of level
         L + 1 — that wave did so while the others worked (after the barrier of ITS last block), and what is left between two
         barriers of the chain is border read -> taps -> arithmetic -> sample write */
      const unsigned long long same = __ballot((int)(comp && lane < nvalid && ((ex.x >> 17) & 3u) == (uint32_t)c));
      /* intra pictures: a 16x16 / 32x32 block is EVERY wave's of its component — each gathers (and smooths) the whole border itself and
         predicts its share of the rows (below): no barrier beyond the level's own, and the block costs the chain about one small
         block's time instead of 2 / 4.5 (the round trip is the border's, the arithmetic is 1 / 4 .. 1 / 8 of the samples) */
      const unsigned long long big = same & __ballot((int)(((ex.x >> 14) & 7u) >= (uint32_t)M355_INTRA_SHARE_MIN_LOG2 && !(ex.x & M355_IBX_PCM)));
      const unsigned long long few = same & ~big;
      const int rank = taken + __popcll(few & ((1ull << lane) - 1ull));
      mine = (few & __ballot((int)((rank & (G - 1)) == g))) | big;
      taken += __popcll(few);
    }
    /* the wave's NEXT block, decoded: record words, offsets of its first sample in the body / residual tiles, mode parameters,
       and (4x4 / 8x8) the plan entries of its border, entry e in lane e */
    int nsrc = -1, nlevel = -1;
    uint32_t e0 = 0, e1 = 0, e3 = 0, ncode = 0;
    int d_bofs = 0, d_angle = 0, d_inv = 0, d_cls = 0, d_log2 = 0, d_small = 0;
    int d_baddr = 0;                                        /* (4x4 / 8x8) this lane's sample in the body */
    auto fetch_next = [&]() {
      nsrc = mine ? __ffsll(mine) - 1 : -1;
      mine &= mine - 1;
      if (nsrc < 0) return;
      e0 = (uint32_t)__builtin_amdgcn_readlane((int)ex.x, nsrc); e1 = (uint32_t)__builtin_amdgcn_readlane((int)ex.y, nsrc);
      const uint32_t e2 = (uint32_t)__builtin_amdgcn_readlane((int)ex.z, nsrc);
      e3 = (uint32_t)__builtin_amdgcn_readlane((int)ex.w, nsrc);
      nlevel = (int)((e3 >> 16) & 0x3FFFu);
      d_log2 = (int)((e0 >> 14) & 7u);
      const int lx = (int)(e0 & 127u), ly = (int)((e0 >> 7) & 127u);
      d_bofs = ly * BODY_PITCH + lx + BODY_X0;
      d_cls = (int)((e2 >> 8) & 7u); d_angle = (int)(int8_t)(e2 & 0xFFu); d_inv = (int)(int16_t)(e2 >> 16);
      d_small = (d_log2 <= 3 && !(e0 & M355_IBX_PCM)) ? 1 : 0;
      if (d_small) {
        ncode = s_plan[((e3 & 0xFFFFu) - plan_lo) + (uint32_t)min(lane, 4 << d_log2)];
        d_baddr = d_bofs + (d_log2 == 2 ? lofs_b4 : lofs_b8);
      }
      M355_PIN_S(d_small); M355_PIN_S(d_bofs); M355_PIN_S(d_cls); M355_PIN_S(d_angle); M355_PIN_S(d_inv); M355_PIN_S(d_log2); M355_PIN_S(nlevel);
    };
    fetch_next();
    bool pend = false;                                       /* the next block is still to be fetched (done behind the barrier) */
#define PROF_T(k) do { } while (0)
    for (int L = lv_first; L <= lv_last; L++) {
    PROF_T(0);
    if (pend) { fetch_next(); pend = false; }
    while (nsrc >= 0 && nlevel == L) {
      const uint32_t code0 = ncode;
      /* first thing behind the barrier: the block's border (fill_from_image + substitution, intrapred.h:534-665, resolved by
         k_intra_plan: one sample per plan entry) and its residual — everything else of the block is decoded beside these reads */
      uint32_t bv0 = 0;
      int rs0 = 0;
      if (__builtin_expect(d_small, 1)) { bv0 = body[code0]; rs0 = (int)(int16_t)body[d_baddr]; }
      const int log2 = d_log2, nT = 1 << log2, cls = d_cls, angle = d_angle, inv = d_inv;
      const int mode = (int)((e0 >> 19) & 63u);
      const bool vert = mode >= 18;
      const bool has_res = (e0 & M355_IBX_HAS_RES) != 0, bfilt = (e0 & M355_IBX_BFILT) != 0;
      const bool pub_col = (e0 & M355_IBX_PUB_COL) != 0, pub_row = (e0 & M355_IBX_PUB_ROW) != 0;
      const int lx = (int)(e0 & 127u), ly = (int)((e0 >> 7) & 127u);      /* (used by the 16x16 / 32x32 / raw paths and by publishing) */

      /* ---- 4x4 / 8x8 (the chain of an intra picture): border entry e in lane e of ONE register, one sample per lane; every
         lane runs the arithmetic (a cross-lane read needs its source lane active), lanes beyond the block do not store ---- */
      auto small_block = [&](auto l2_) {
        constexpr int LOG2 = decltype(l2_)::value, NT = 1 << LOG2, Z = 2 * NT, NENT = 4 * NT + 1;
        const int x = lane & (NT - 1), y = lane >> LOG2;
        const bool inb = lane < NT * NT;
        uint32_t bv = bv0;
        int rs = 0;
        /* (branch-free where it is cheap: a taken branch costs a lone wave more than the few instructions it skips) */
        rs = rs0 & (has_res ? -1 : 0);
        if (__builtin_expect(__any((int)(bv == HALO_NOT_READY)), 0)) {
          /* a halo sample its CTB has not published yet: poll its granule */
          const bool pending = bv == HALO_NOT_READY && code0 >= (uint32_t)HALO_BASE && code0 < (uint32_t)(HALO_BASE + HALO_N);
          if (__any((int)pending))
            bv = d_poll_halo(d_edge_row(p, cs, ctbY - 1, 0), d_edge_col(p, cs, ctbX - 1, 0), p.timeout, halo, (int)code0 - HALO_BASE, bv, pending, x0c, y0c, epoch);
        }
        PROF_T(1);
        {   /* intra_prediction_sample_filtering (intrapred.h:185-258), [1 2 1] only (strong smoothing is 32x32) */
          const uint32_t nb = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)bv, 0x138, 0xF, 0xF, false) + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)bv, 0x130, 0xF, 0xF, false);
          const uint32_t fbv = (nb + 2u * bv + 2u) >> 2;
          if ((e0 & M355_IBX_FILT) && lane > 0 && lane < NENT - 1) bv = fbv;
        }
#define BRL(i) ((int)__builtin_amdgcn_readlane((int)bv, (i) + Z))                 /* border entry i, the same for every lane */
#define BRP(i) ((int)__builtin_amdgcn_ds_bpermute(((i) + Z) << 2, (int)bv))       /* border entry i, per lane */
        int v;
        if (__builtin_expect(cls >= 3, 1)) {
          /* angular (33 of the 35 modes; first in line, and one formula for both signs of the angle): ref[] of the reference is not
             built — its entry i is border entry sgn*i for i >= 0 and, left of the corner (negative angles only), -sgn*((i*invAngle+128)>>8) */
          const int a_ = vert ? y : x, b_ = vert ? x : y;
          const int tt = __mul24(a_ + 1, angle), iIdx = tt >> 5, iFact = tt & 31;
          const int x1 = b_ + iIdx + 1, x2 = x1 + 1;
          const int p1 = (__mul24(x1, inv) + 128) >> 8, p2 = (__mul24(x2, inv) + 128) >> 8;     /* (invAngle is 0 for a positive angle, whose x1 is never negative) */
          int i1 = x1 >= 0 ? x1 : -p1, i2 = x2 >= 0 ? x2 : -p2;
          if (!vert) { i1 = -i1; i2 = -i2; }
          const int r1 = BRP(i1), r2 = BRP(i2);
          v = (32 * r1 + __mul24(iFact, r2 - r1) + 16) >> 5;   /* (= r1 when iFact is 0) */
        } else if (cls == 0) {        /* planar (intrapred.h:261-285) */
          const int l = BRP(-1 - y), t = BRP(1 + x), tr = BRL(1 + NT), bl = BRL(-1 - NT);
          v = (__mul24(NT - 1 - x, l) + __mul24(x + 1, tr) + __mul24(NT - 1 - y, t) + __mul24(y + 1, bl) + NT) >> (LOG2 + 1);   /* (24-bit multiplies: full rate) */
        } else if (cls == 1) {        /* DC (intrapred.h:288-310, 378-392) */
          int s_ = (lane >= Z - NT && lane <= Z + NT && lane != Z) ? (int)bv : 0;
          s_ += __builtin_amdgcn_update_dpp(0, s_, 0xB1, 0xF, 0xF, false);     /* quad_perm [1,0,3,2] */
          s_ += __builtin_amdgcn_update_dpp(0, s_, 0x4E, 0xF, 0xF, false);     /* quad_perm [2,3,0,1] */
          s_ += __builtin_amdgcn_update_dpp(0, s_, 0x141, 0xF, 0xF, false);    /* row_half_mirror */
          s_ += __builtin_amdgcn_update_dpp(0, s_, 0x140, 0xF, 0xF, false);    /* row_mirror: every lane of a row of 16 holds the row's sum */
          int sum = __builtin_amdgcn_readlane(s_, 0) + __builtin_amdgcn_readlane(s_, 16);
          const int dc = (sum + NT) >> (LOG2 + 1);
          v = dc;
          if (bfilt) {
            const int eb = BRP(y == 0 ? x + 1 : -y - 1), e2c = BRL(-1) + BRL(1);
            if (x == 0 || y == 0) v = (eb + 3 * dc + 2) >> 2;
            if (lane == 0) v = (e2c + 2 * dc + 2) >> 2;
          }
        } else {                      /* pure horizontal / vertical (intrapred.h:330-433 with intraPredAngle 0) */
          const int l = BRP(-1 - y), t = BRP(1 + x), corner = BRL(0), first = vert ? BRL(1) : BRL(-1);
          v = vert ? t : l;
          if (bfilt && (vert ? x == 0 : y == 0)) v = d_clip3(0, pix_max, first + (((vert ? l : t) - corner) >> 1));
        }
#undef BRL
#undef BRP
        v = d_clip3(0, pix_max, v + rs);   /* (a prediction is inside the sample range: no-op without a residual) */
        if (inb) body[d_baddr] = (uint16_t)v;
        /* ---- publish from the registers: a granule = two samples, the second one comes from the lane below / beside ---- */
        if (__builtin_expect((e0 & (M355_IBX_PUB_COL | M355_IBX_PUB_ROW)) != 0, 0)) {
        if (pub_col) {
          const uint32_t v2 = (uint32_t)__builtin_amdgcn_ds_bpermute((lane + NT) << 2, v);
          if (inb && x == NT - 1 && !(y & 1))
            __hip_atomic_store(d_edge_col(p, cs, ctbX, y0c + ly + y), ((m355_granule)epoch << 32) | (v2 << 16) | (uint32_t)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (pub_row) {
          const uint32_t v2 = (uint32_t)__builtin_amdgcn_ds_bpermute((lane + 1) << 2, v);
          if (inb && y == NT - 1 && !(x & 1))
            __hip_atomic_store(d_edge_row(p, cs, ctbY, x0c + lx + x), ((m355_granule)epoch << 32) | (v2 << 16) | (uint32_t)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        }
      };

      if (__builtin_expect(d_small, 1)) {
        if (log2 == 2) small_block(std::integral_constant<int, 2>()); else small_block(std::integral_constant<int, 3>());
      } else {
      /* this wave's share of a 16x16 / 32x32 block: passes (64 samples = 4 / 2 rows each) [it_lo, it_hi) of the block's 4 / 16 */
      const int nPass = (nT * nT) >> 6;
      const bool shared = log2 >= M355_INTRA_SHARE_MIN_LOG2;
      const int it_lo = shared ? (nPass * g) / G : 0, it_hi = shared ? (nPass * (g + 1)) / G : nPass;
      if (!(e0 & M355_IBX_PCM)) {
        /* ---- 16x16 / 32x32: the border lives in LDS (65 / 129 entries) ---- */
        if (it_lo < it_hi) {
        const int nEnt = 4 * nT + 1, Z = 2 * nT;
        const uint16_t* pl = s_plan + ((e3 & 0xFFFFu) - plan_lo);
        /* (all plan entries, then all samples: two LDS round trips for the whole border, not two per 64-entry chunk) */
        uint32_t cd[3] = {0, 0, 0}, val[3] = {0, 0, 0};
#pragma unroll
        for (int q = 0; q < 3; q++) {
          if (64 * q >= nEnt) continue;              /* wave-uniform */
          const int e = lane + 64 * q;
          if (e < nEnt) cd[q] = (uint32_t)pl[e];
        }
#pragma unroll
        for (int q = 0; q < 3; q++) {
          if (64 * q >= nEnt) continue;
          if (lane + 64 * q < nEnt) val[q] = body[cd[q]];
        }
        if (__any((int)(val[0] == HALO_NOT_READY || val[1] == HALO_NOT_READY || val[2] == HALO_NOT_READY))) {
#pragma unroll
          for (int q = 0; q < 3; q++) {
            if (64 * q >= nEnt) continue;
            const bool pending = lane + 64 * q < nEnt && val[q] == HALO_NOT_READY && cd[q] >= (uint32_t)HALO_BASE && cd[q] < (uint32_t)(HALO_BASE + HALO_N);
            if (__any((int)pending))
              val[q] = d_poll_halo(d_edge_row(p, cs, ctbY - 1, 0), d_edge_col(p, cs, ctbX - 1, 0), p.timeout, halo, (int)cd[q] - HALO_BASE, val[q], pending, x0c, y0c, epoch);
          }
        }
#pragma unroll
        for (int q = 0; q < 3; q++) {
          if (64 * q >= nEnt) continue;
          if (lane + 64 * q < nEnt) raw[lane + 64 * q] = (uint16_t)val[q];
        }
        wave_sync();
        PROF_T(1);
        uint16_t* P = raw; /* border in use, entry index = i + 2nT */
        if (e0 & M355_IBX_FILT) {     /* intra_prediction_sample_filtering (intrapred.h:185-258) */
          const bool bi = (e0 & M355_IBX_STRONG) && d_abs((int)raw[Z] + raw[Z + 64] - 2 * raw[Z + 32]) < thr_strong && d_abs((int)raw[Z] + raw[Z - 64] - 2 * raw[Z - 32]) < thr_strong;
#pragma unroll
          for (int q = 0; q < 3; q++) {
            if (64 * q >= nEnt) continue;
            const int e = lane + 64 * q;
            if (e < nEnt) {
              const int i = e - Z;
              int v;
              if (i == -Z || i == Z) v = raw[e];
              else if (bi) {
                if (i == 0) v = raw[Z];
                else if (i < 0) v = raw[Z] + (((-i) * ((int)raw[Z - 64] - raw[Z]) + 32) >> 6);
                else v = raw[Z] + ((i * ((int)raw[Z + 64] - raw[Z]) + 32) >> 6);
              } else v = (raw[e + 1] + 2 * raw[e] + raw[e - 1] + 2) >> 2;
              pf[e] = (uint16_t)v;
            }
          }
          wave_sync();
          P = pf;
        }
#define BRD(i) ((int)P[(i) + Z])
        int dcVal = 0;
        if (mode == 1) {
          int s_ = 0;
          if (lane < nT) s_ = BRD(lane + 1) + BRD(-lane - 1);
          s_ += __builtin_amdgcn_update_dpp(0, s_, 0xB1, 0xF, 0xF, false);     /* sums over rows of 16 lanes, as in the 4x4 / 8x8 path */
          s_ += __builtin_amdgcn_update_dpp(0, s_, 0x4E, 0xF, 0xF, false);
          s_ += __builtin_amdgcn_update_dpp(0, s_, 0x141, 0xF, 0xF, false);
          s_ += __builtin_amdgcn_update_dpp(0, s_, 0x140, 0xF, 0xF, false);
          dcVal = (__builtin_amdgcn_readlane(s_, 0) + __builtin_amdgcn_readlane(s_, 16) + nT) >> (log2 + 1);
        }
        {
          /* one loop per mode class, FOUR samples per lane in flight: their border taps and residuals are requested together (one
             LDS round trip per four samples, not two per sample), then the four are computed and stored */
          const int xb = lane & (nT - 1), yb = lane >> log2, ystep = 64 >> log2, nIt = (nT * nT) >> 6;
          const int res_on = has_res ? -1 : 0;
          auto big_loop = [&](auto cls_) {
            constexpr int CLS = decltype(cls_)::value;
            const int x = xb;
            /* what does not depend on the row */
            const int t_x = (CLS == 0 || CLS == 2) ? BRD(1 + x) : 0;
            const int u_tr = CLS == 0 ? BRD(1 + nT) : 0, u_bl = CLS == 0 ? BRD(-1 - nT) : 0;
            const int u_c = CLS == 2 ? BRD(0) : 0, u_f = CLS == 2 ? (vert ? BRD(1) : BRD(-1)) : 0;
            const int u_e2 = CLS == 1 ? BRD(-1) + BRD(1) : 0;
            (void)nIt;
            for (int it0 = it_lo; it0 < it_hi; it0 += 4) {
              int ta[4], tb[4], fa[4], rs[4];
#pragma unroll
              for (int u = 0; u < 4; u++) {
                const int y = yb + min(it0 + u, it_hi - 1) * ystep;      /* (a share of 1 or 2 passes: the rest repeats the last one) */
                fa[u] = 0; tb[u] = 0;
                if (CLS == 0 || CLS == 2) ta[u] = BRD(-1 - y);
                else if (CLS == 1) ta[u] = BRD(y == 0 ? x + 1 : -y - 1);
                else {
                  const int a_ = vert ? y : x, b_ = vert ? x : y;
                  const int tt = __mul24(a_ + 1, angle), iIdx = tt >> 5;
                  fa[u] = tt & 31;
                  const int x1 = b_ + iIdx + 1, x2 = x1 + 1;
                  int i1, i2;
                  if (CLS == 3) { i1 = vert ? x1 : -x1; i2 = vert ? x2 : -x2; }
                  else {
                    const int p1 = (__mul24(x1, inv) + 128) >> 8, p2 = (__mul24(x2, inv) + 128) >> 8;
                    i1 = x1 >= 0 ? x1 : -p1; i2 = x2 >= 0 ? x2 : -p2;
                    if (!vert) { i1 = -i1; i2 = -i2; }
                  }
                  ta[u] = BRD(i1); tb[u] = BRD(fa[u] ? i2 : i1);   /* (no read beyond the border when the second tap has weight 0) */
                }
                rs[u] = (int)(int16_t)body[d_bofs + y * BODY_PITCH + x] & res_on;
              }
#pragma unroll
              for (int u = 0; u < 4; u++) {
                const int y = yb + min(it0 + u, it_hi - 1) * ystep;
                int v;
                if (CLS == 0) v = (__mul24(nT - 1 - x, ta[u]) + __mul24(x + 1, u_tr) + __mul24(nT - 1 - y, t_x) + __mul24(y + 1, u_bl) + nT) >> (log2 + 1);
                else if (CLS == 1) {
                  v = dcVal;
                  if (bfilt) {
                    if (x == 0 || y == 0) v = (ta[u] + 3 * dcVal + 2) >> 2;
                    if (x == 0 && y == 0) v = (u_e2 + 2 * dcVal + 2) >> 2;
                  }
                } else if (CLS == 2) {
                  v = vert ? t_x : ta[u];
                  if (bfilt && (vert ? x == 0 : y == 0)) v = d_clip3(0, pix_max, u_f + (((vert ? ta[u] : t_x) - u_c) >> 1));
                } else v = (32 * ta[u] + __mul24(fa[u], tb[u] - ta[u]) + 16) >> 5;
                v = d_clip3(0, pix_max, v + rs[u]);          /* (a prediction is inside the sample range: no-op without a residual) */
                body[d_bofs + y * BODY_PITCH + x] = (uint16_t)v;         /* for the next blocks' borders; the picture is written at the end */
              }
            }
          };
          if (cls == 0) big_loop(std::integral_constant<int, 0>());
          else if (cls == 1) big_loop(std::integral_constant<int, 1>());
          else if (cls == 2) big_loop(std::integral_constant<int, 2>());
          else if (cls == 3) big_loop(std::integral_constant<int, 3>());
          else big_loop(std::integral_constant<int, 4>());
        }
#undef BRD
        }   /* a share of the block */
      } else { /* raw block (slice.cc:4211-4255) */
        for (int o = lane; o < nT * nT; o += 64) {
          const int y = o >> log2, x = o & (nT - 1);
          body[(ly + y) * BODY_PITCH + lx + x + BODY_X0] = p.pcm[e1 + o];
        }
      }
      /* ---- publish: the block's share of the CTB's right column / bottom row, two samples per granule (of a shared block: the rows
         this wave predicted — a pass is 4 / 2 whole rows —, the bottom row by the wave that holds the last pass) ---- */
      const int pr_lo = (e0 & M355_IBX_PCM) ? 0 : it_lo * (64 >> log2), pr_hi = (e0 & M355_IBX_PCM) ? nT : it_hi * (64 >> log2);
      if (pub_col || (pub_row && pr_hi == nT)) {
        wave_sync();
        if (pub_col && lane < (nT >> 1) && 2 * lane >= pr_lo && 2 * lane < pr_hi) {
          const int y = ly + 2 * lane;
          const uint32_t s0 = body[y * BODY_PITCH + lx + nT - 1 + BODY_X0], s1 = body[(y + 1) * BODY_PITCH + lx + nT - 1 + BODY_X0];
          __hip_atomic_store(d_edge_col(p, cs, ctbX, y0c + y), ((m355_granule)epoch << 32) | (s1 << 16) | s0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (pub_row && pr_hi == nT && lane >= 32 && lane < 32 + (nT >> 1)) {
          const int x = lx + 2 * (lane - 32);
          const uint32_t s0 = body[(ly + nT - 1) * BODY_PITCH + x + BODY_X0], s1 = body[(ly + nT - 1) * BODY_PITCH + x + 1 + BODY_X0];
          __hip_atomic_store(d_edge_row(p, cs, ctbY, x0c + x), ((m355_granule)epoch << 32) | (s1 << 16) | s0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      }   /* 16x16 / 32x32 / raw */
      PROF_T(2);
      /* another block of this level?  Else go to the barrier first: the next record is fetched behind it */
      const int peek = mine ? __builtin_amdgcn_readlane(lv, __ffsll(mine) - 1) : -1;
      if (peek != L) { pend = true; break; }
      fetch_next();
    }   /* this wave's blocks of the level */
    PROF_T(3);
    SYNC_CTB();
    PROF_T(4);
    /* level done: its samples are in LDS for the next level's borders (one wave per component: its own blocks are ordered
       by the wave-level sync; components do not interact) */
    }   /* levels in the batch */
  }   /* 64-record batches */
  /* ---- the CTB's intra samples -> the picture: every 4x4 unit some intra block covered (s_cover), one 4-sample row piece per
     lane, neighbouring lanes on neighbouring pieces of a row (the last level's barrier / wave_sync made them all visible) ---- */
  if (comp) {
    const int l2u = (l2c - csw) - 2;                         /* log2(units per row) */
    const int npieces = ch << l2u;
    for (int idx = lane + 64 * g; idx < npieces; idx += 64 * G) {
      const int y = idx >> l2u, u = idx & ((1 << l2u) - 1);
      if (!((s_cover[cs][y >> 2] >> u) & 1u)) continue;
      const uint2 v = *(const uint2*)(body + y * BODY_PITCH + BODY_X0 + 4 * u);
      PIX* dst = plane + (size_t)(y0c + y) * stride + x0c + 4 * u;
      if (sizeof(PIX) == 2) *(uint2*)dst = v;
      else *(uint32_t*)dst = (v.x & 0xFFu) | ((v.x >> 8) & 0xFF00u) | ((v.y & 0xFFu) << 16) | ((v.y & 0xFF0000u) << 8);
    }
  }
  if (!DENSE) return;
  __syncthreads();     /* the LDS tiles (and the ticket word) are free for the workgroup's next CTB */
  }   /* persistent workgroup: next CTB */
#undef SYNC_CTB
}

#ifndef M355_INTRA_DENSE_NW
#define M355_INTRA_DENSE_NW 12
#endif
#ifndef M355_INTRA_SPARSE_NW
#define M355_INTRA_SPARSE_NW 4    /* inter pictures: up to 2 luma + 1 + 1 chroma waves per CTB */
#endif

template <class PIX, int CF>
static void launch_intra_cf(const DevPic& p, bool ticket_zero, hipStream_t st)
{
  if (!ticket_zero) hipMemsetAsync(p.ticket, 0, 4, st);      /* (an inter picture's k_job_count zeroes it: one packet less) */
  /* dense intra pictures: 12 waves (up to 8 luma + 2 + 2 chroma blocks of a level at once); sparse ones: 4 (3 and 6 measured
     slower) */
  const dim3 dense_grid(p.intra_grid > 0 && p.intra_grid < p.n_intra_work ? p.intra_grid : p.n_intra_work);
  if (p.intra_dense && p.intra_keeper) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_intra<PIX, CF, M355_INTRA_KEEPER_NW, true, false>), dense_grid, dim3(64 * M355_INTRA_KEEPER_NW), 0, st, p, p.n_intra_work, (const DevPic*)nullptr, 1, (uint32_t*)nullptr);
  else if (p.intra_dense) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_intra<PIX, CF, M355_INTRA_DENSE_NW, true, false>), dense_grid, dim3(64 * M355_INTRA_DENSE_NW), 0, st, p, p.n_intra_work, (const DevPic*)nullptr, 1, (uint32_t*)nullptr);
  else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_intra<PIX, CF, M355_INTRA_SPARSE_NW, false, false>), dim3(p.n_intra_work), dim3(64 * M355_INTRA_SPARSE_NW), 0, st, p, p.n_intra_work, (const DevPic*)nullptr, 1, (uint32_t*)nullptr);
}

void m355_launch_intra(const DevPic& p, bool hbd, hipStream_t st, bool ticket_zero)
{
  if (!p.n_intra_work) return;
  switch (p.pp.chroma_format_idc) {
    case 0: if (hbd) launch_intra_cf<uint16_t, 0>(p, ticket_zero, st); else launch_intra_cf<uint8_t, 0>(p, ticket_zero, st); break;
    case 1: if (hbd) launch_intra_cf<uint16_t, 1>(p, ticket_zero, st); else launch_intra_cf<uint8_t, 1>(p, ticket_zero, st); break;
    case 2: if (hbd) launch_intra_cf<uint16_t, 2>(p, ticket_zero, st); else launch_intra_cf<uint8_t, 2>(p, ticket_zero, st); break;
    default: if (hbd) launch_intra_cf<uint16_t, 3>(p, ticket_zero, st); else launch_intra_cf<uint8_t, 3>(p, ticket_zero, st); break;
  }
}

/* several intra pictures of one geometry in ONE launch (k_intra<BATCH>): first = the pictures' common parameters, dev_pics = their
   DevPic records in device memory, ticket = a zeroed word of its own, grid = persistent workgroups */
template <class PIX, int CF>
static void launch_intra_batch_cf(const DevPic& first, const DevPic* dev_pics, int n, int max_work, uint32_t* ticket, int grid, hipStream_t st)
{
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k_intra<PIX, CF, M355_INTRA_DENSE_NW, true, true>), dim3(grid), dim3(64 * M355_INTRA_DENSE_NW), 0, st, first, max_work, dev_pics, n, ticket);
}
void m355_launch_intra_batch(const DevPic& first, bool hbd, const DevPic* dev_pics, int n, int max_work, uint32_t* ticket, int grid, hipStream_t st)
{
  if (max_work <= 0 || n <= 0) return;
  switch (first.pp.chroma_format_idc) {
    case 0: if (hbd) launch_intra_batch_cf<uint16_t, 0>(first, dev_pics, n, max_work, ticket, grid, st); else launch_intra_batch_cf<uint8_t, 0>(first, dev_pics, n, max_work, ticket, grid, st); break;
    case 1: if (hbd) launch_intra_batch_cf<uint16_t, 1>(first, dev_pics, n, max_work, ticket, grid, st); else launch_intra_batch_cf<uint8_t, 1>(first, dev_pics, n, max_work, ticket, grid, st); break;
    case 2: if (hbd) launch_intra_batch_cf<uint16_t, 2>(first, dev_pics, n, max_work, ticket, grid, st); else launch_intra_batch_cf<uint8_t, 2>(first, dev_pics, n, max_work, ticket, grid, st); break;
    default: if (hbd) launch_intra_batch_cf<uint16_t, 3>(first, dev_pics, n, max_work, ticket, grid, st); else launch_intra_batch_cf<uint8_t, 3>(first, dev_pics, n, max_work, ticket, grid, st); break;
  }
}

void m355_launch_intra_plan_batch(const HostBatch& b, hipStream_t st)
{
  int work = 0, cf = 1;
  for (int k = 0; k < b.n; k++) if ((b.on >> k) & 1u) { work = std::max(work, b.host[k].n_intra_work); cf = b.host[k].pp.chroma_format_idc; }
  if (!work) return;
  const DevBatch d{b.dev, b.on};
  const dim3 grid(work, PLAN_SPLIT, b.n);
  switch (cf) {
    case 0: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_intra_plan_batch<0>), grid, dim3(256), 0, st, d); break;
    case 1: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_intra_plan_batch<1>), grid, dim3(256), 0, st, d); break;
    case 2: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_intra_plan_batch<2>), grid, dim3(256), 0, st, d); break;
    default: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_intra_plan_batch<3>), grid, dim3(256), 0, st, d); break;
  }
}

/* the plans of all intra blocks of the picture (needs the CU plane only under constrained intra prediction: launched behind
   the metadata planes on the side stream, beside k_inter / k_residual) */
void m355_launch_tu_plan(const DevPic& p, hipStream_t st)
{
  /* (one intra picture at a time: its CTBs are planned by k_intra itself, the launch carries the transform edges only) */
  const int n_plan = (p.intra_dense && p.intra_keeper) ? 0 : p.n_intra_work, split = p.intra_dense ? PLAN_SPLIT : 1;
  const int nb_tu = (p.n_tus + 255) / 256, nb = nb_tu + n_plan * split;
  if (!nb) return;
  switch (p.pp.chroma_format_idc) {
    case 0: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_tu_plan<0>), dim3(nb), dim3(256), 0, st, p, nb_tu, n_plan, split); break;
    case 1: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_tu_plan<1>), dim3(nb), dim3(256), 0, st, p, nb_tu, n_plan, split); break;
    case 2: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_tu_plan<2>), dim3(nb), dim3(256), 0, st, p, nb_tu, n_plan, split); break;
    default: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_tu_plan<3>), dim3(nb), dim3(256), 0, st, p, nb_tu, n_plan, split); break;
  }
}

void m355_launch_intra_plan(const DevPic& p, hipStream_t st)
{
  if (!p.n_intra_work) return;
  if (p.intra_dense && p.intra_keeper) return;             /* planned by k_intra itself */
  /* workgroups per CTB: PLAN_SPLIT for an intra picture (hundreds of blocks per CTB); ONE for the handful of intra blocks a CTB of an
     inter picture holds (3.4 on average at C5: with eight workgroups 90 000 waves were launched for 28 000 blocks) */
  const int split = p.intra_dense ? PLAN_SPLIT : 1;
  switch (p.pp.chroma_format_idc) {
    case 0: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_intra_plan<0>), dim3(p.n_intra_work, split), dim3(256), 0, st, p, p.n_intra_work); break;
    case 1: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_intra_plan<1>), dim3(p.n_intra_work, split), dim3(256), 0, st, p, p.n_intra_work); break;
    case 2: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_intra_plan<2>), dim3(p.n_intra_work, split), dim3(256), 0, st, p, p.n_intra_work); break;
    default: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_intra_plan<3>), dim3(p.n_intra_work, split), dim3(256), 0, st, p, p.n_intra_work); break;
  }
}
