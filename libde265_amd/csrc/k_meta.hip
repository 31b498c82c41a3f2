/*
 * k_meta.hip — rasterise the CU / transform-leaf / PB lists into the metadata planes the in-loop
 * filters and the intra availability tests read.  This replaces, on the device, what the reference
 * keeps incrementally in de265_image::cb_info / tu_info / pb_info / deblk_info (image.h:389-395)
 * and derive_edgeFlags_CTBRow + markTransformBlockBoundary + markPredictionBlockBoundary
 * (deblock.cc:33-227).  Pure scatter kernels: HBM-write bound, a few bytes per 4x4 unit.
 */
#include <algorithm>
#include "k_common.h"
#include "k_meta_tu.h"

/* one thread per CU: CU-index plane, per-CU edge decisions (deblock.cc:172-210), PB edges */
__device__ __forceinline__ void k_meta_cu_body(const DevPic& p, const int blk)
{
  M355_GATE(p);
  const int i = blk * blockDim.x + threadIdx.x;
  if (i >= p.n_cus) return;
  const m355_cu cu = p.cus[i];
  const int x0 = cu.x, y0 = cu.y;
  const int l2m = p.pp.log2_min_cb_size;
  const int n = 1 << (cu.log2_size - l2m);
  const int cx = x0 >> l2m, cy = y0 >> l2m;
  for (int y = cy; y < cy + n && y < p.hcb; y++)
    for (int x = cx; x < cx + n && x < p.wcb; x++) p.cb_cu[y * p.wcb + x] = (uint32_t)i + 1;

  const int ctb_mask = (1 << p.pp.log2_ctb_size) - 1;
  const m355_slice sh = d_slice_at(p, x0, y0);
  int left = x0 != 0, top = y0 != 0;
  if (x0 && (x0 & ctb_mask) == 0) {
    if (!(sh.flags & M355_SF_LF_ACROSS_SLICES) && sh.slice_addr_rs != d_slice_at(p, x0 - 1, y0).slice_addr_rs) left = 0;
    else if (!(p.pp.flags & M355_PF_LF_ACROSS_TILES) && p.tile_id[d_ctb_of(p, x0, y0)] != p.tile_id[d_ctb_of(p, x0 - 1, y0)]) left = 0;
  }
  if (y0 && (y0 & ctb_mask) == 0) {
    if (!(sh.flags & M355_SF_LF_ACROSS_SLICES) && sh.slice_addr_rs != d_slice_at(p, x0, y0 - 1).slice_addr_rs) top = 0;
    else if (!(p.pp.flags & M355_PF_LF_ACROSS_TILES) && p.tile_id[d_ctb_of(p, x0, y0)] != p.tile_id[d_ctb_of(p, x0, y0 - 1)]) top = 0;
  }
  const int en = !(sh.flags & M355_SF_DEBLOCK_DISABLED);
  p.cuf[i] = (uint8_t)(left | (top << 1) | (en << 2));
  if (!en) return;

  /* markPredictionBlockBoundary (deblock.cc:68-129) */
  const int cb = 1 << cu.log2_size, h2 = cb >> 1, q4 = cb >> 2;
  int vx = -1, hy = -1;
  switch (cu.part_mode) {
    case 3: vx = h2; hy = h2; break;
    case 2: vx = h2; break;
    case 1: hy = h2; break;
    case 6: vx = q4; break;
    case 7: vx = h2 + q4; break;
    case 4: hy = q4; break;
    case 5: hy = h2 + q4; break;
    default: break;
  }
  /* a unit may receive both bits (NxN centre): the vertical pass writes first, the horizontal ORs */
  if (vx >= 0) {
    const int ux = (x0 + vx) >> 2;
    if (ux < p.w4)
      for (int k = 0; k < cb; k += 4) {
        const int uy = (y0 + k) >> 2;
        if (uy < p.h4) p.edge_pb[uy * p.w4 + ux] = E_PB_V;
      }
  }
  if (hy >= 0) {
    const int uy = (y0 + hy) >> 2;
    if (uy < p.h4)
      for (int k = 0; k < cb; k += 4) {
        const int ux = (x0 + k) >> 2;
        if (ux < p.w4) p.edge_pb[uy * p.w4 + ux] |= E_PB_H;
      }
  }
}

/* ---- job counts on the device (the host no longer reads the PB list of a picture recorded in place) ----
 * k_job_count: one workgroup per 256-PB chunk (= one k_meta_pb workgroup): the chunk's jobs per range -> job_base[chunk][3].
 * k_meta_pb  : every workgroup sums the counts of the chunks in front of its own (and all of them, for the range starts: one list, two
 *              lists, picture edge — one after the other) itself: 3 x n_chunks words out of the L2 per workgroup instead of a
 *              one-workgroup scan launch on the picture's critical path; workgroup 0 leaves the range ends in job_tot[0..2].  A malformed record counts nothing here (k_validate
 *              rejects the picture); lists whose blocks overlap can exceed jobs[]: the ends are clamped and k_meta_pb drops what
 *              does not fit (memory-safe, the picture is garbage either way). */
__device__ __forceinline__ int d_pb_jobs(const m355_pb& pb)
{
  if (pb.w < 4 || pb.h < 4 || pb.w > 64 || pb.h > 64 || (pb.w & 3) || (pb.h & 3)) return 0;
  return (pb.w >> 2) * ((pb.h + 7) >> 3);
}
/* job class of a PB (= the range of the job list its jobs go to, k_inter_jobs): 0 one list, 1 two lists, 2 explicit weights,
   3 EDGE = a reference window leaves the picture (coordinate-clamped loads) */
__device__ __forceinline__ int d_pb_class(const DevPic& p, const m355_pb& pb)
{
  if (m355_pb_is_edge(pb, p.pp.width, p.pp.height, p.pp.chroma_format_idc)) return 3;
  if (pb.flags & M355_PBF_WEIGHTED) return 2;
  return ((pb.flags & M355_PBF_MC_L0) && (pb.flags & M355_PBF_MC_L1)) ? 1 : 0;
}
/* 16-byte units of edge_tu | edge_pb | cb_cu (prepare() in runtime_decode.hip: cb_cu starts at the next multiple of 64, 64 spare bytes behind) */
__host__ __device__ static inline size_t d_meta_fill16(const DevPic& p) { return ((((size_t)2 * p.w4 * p.h4 + 63) & ~(size_t)63) + (size_t)p.wcb * p.hcb * 4 + 15) / 16; }
/* clear_planes: the launch also zero-fills the metadata planes (edge_tu | edge_pb | cb_cu) that k_meta_planes scatters into — it is the
   first kernel of an inter picture on its lane's main stream, in FRONT of the fork of the side stream: one launch less per picture
   than a fill of its own — and resets the ticket word of k_intra, another 4-byte fill packet (a packet costs about 2 us of pipeline time, profiles/r04_aj_*) */
__global__ void __launch_bounds__(256) k_job_count(DevPic p, int clear_planes)
{
  M355_GATE(p);
  if (clear_planes) {
    if (blockIdx.x == 0 && threadIdx.x == 0) *p.ticket = 0;   /* k_intra's claim counter (the lane's previous k_intra is over: stream order) */
    const size_t n16 = d_meta_fill16(p);
    uint4* q = (uint4*)p.edge_tu;
    for (size_t k = (size_t)blockIdx.x * 256 + threadIdx.x; k < n16; k += (size_t)gridDim.x * 256) q[k] = make_uint4(0, 0, 0, 0);
  }
  const int i = blockIdx.x * 256 + threadIdx.x;
  int cnt[4] = {0, 0, 0, 0};
  if (i < p.n_pbs) {
    const m355_pb pb = p.pbs[i];
    const int cls = d_pb_class(p, pb);
    const int n = d_pb_jobs(pb);
#pragma unroll
    for (int k = 0; k < 4; k++) cnt[k] = cls == k ? n : 0;
  }
  __shared__ int s_w[4][4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) cnt[k] += __shfl_xor(cnt[k], d, 64);
  }
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int k = 0; k < 4; k++) s_w[threadIdx.x >> 6][k] = cnt[k];
  }
  __syncthreads();
  if (threadIdx.x < 4) p.job_base[blockIdx.x * 4 + threadIdx.x] = (uint32_t)(s_w[0][threadIdx.x] + s_w[1][threadIdx.x] + s_w[2][threadIdx.x] + s_w[3][threadIdx.x]);
}
/* one thread per prediction block: PB index plane (pb_info, image.cc set_mv_info) and the job list of
 * k_inter_jobs: (w/4) x ceil(h/8) jobs per PB, row block major so consecutive jobs are horizontally
 * adjacent; three ranges (one-list, bi-predicted, picture-edge jobs), each in PB order. */
__device__ __forceinline__ void k_meta_pb_body(const DevPic& p, const int chunk, const int n_chunks)
{
  const int i = chunk * 256 + (int)threadIdx.x;
  const int lane = threadIdx.x & 63;
  const bool active = i < p.n_pbs;
  m355_pb pb;
  if (active) pb = p.pbs[i]; else { pb.x = pb.y = 0; pb.w = pb.h = 0; pb.flags = 0; }
  const int ns = pb.w >> 2;
  const int njobs = active ? d_pb_jobs(pb) : 0;              /* (as k_job_count counted it) */
  /* four ranges: one-list jobs, bi-predicted jobs (so a wave never idles through a second pass it does not need), explicitly weighted
     jobs, edge jobs; wave-level exclusive scans */
  const int cls = !active ? 4 : d_pb_class(p, pb);
  int incl[4];
#pragma unroll
  for (int k = 0; k < 4; k++) incl[k] = cls == k ? njobs : 0;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
#pragma unroll
    for (int k = 0; k < 4; k++) { const int t = __shfl_up(incl[k], (unsigned)d, 64); if (lane >= d) incl[k] += t; }
  }
  /* first job of this wave per range = the chunk's base + the totals of the workgroup's earlier waves (no atomics: the job list is
     in PB order, deterministic and spatially coherent) */
  __shared__ int s_wtot[4][4];
  const int wave = threadIdx.x >> 6;
  if (lane == 63) {
#pragma unroll
    for (int k = 0; k < 4; k++) s_wtot[wave][k] = incl[k];
  }
  __syncthreads();
  /* the chunk's first job per range: sum of the counts (k_job_count) of the chunks before it + where the range starts */
  __shared__ uint32_t s_red[4][8];
  {
    uint32_t acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};              /* [0..3] all chunks, [4..7] the chunks in front of this one */
    for (int ch = (int)threadIdx.x; ch < n_chunks; ch += 256) {
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const uint32_t n = p.job_base[ch * 4 + k];
        acc[k] += n;
        acc[4 + k] += ch < chunk ? n : 0u;
      }
    }
#pragma unroll
    for (int k = 0; k < 8; k++) {
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) acc[k] += (uint32_t)__shfl_xor((int)acc[k], d, 64);
    }
    if (lane == 0) {
#pragma unroll
      for (int k = 0; k < 8; k++) s_red[wave][k] = acc[k];
    }
  }
  __syncthreads();
  uint32_t base[4];
  {
    uint32_t tot[8];
#pragma unroll
    for (int k = 0; k < 8; k++) tot[k] = s_red[0][k] + s_red[1][k] + s_red[2][k] + s_red[3][k];
    uint32_t start = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) { base[k] = start + tot[4 + k]; start += tot[k]; }
    if (chunk == 0 && threadIdx.x == 0) {
      const uint32_t cap = p.jobs_cap;
      uint32_t end = 0;
#pragma unroll
      for (int k = 0; k < 4; k++) { end += tot[k]; p.job_tot[k] = min(end, cap); }
    }
#pragma unroll
    for (int k = 0; k < 4; k++)
      for (int w = 0; w < wave; w++) base[k] += (uint32_t)s_wtot[w][k];
  }
  /* emit the wave's jobs cooperatively: slot t of the wave's total belongs to the PB found by a binary
     search over the wave's exclusive scan (shuffles), so a 64x64 PB (128 jobs) costs the wave two
     iterations instead of stalling 63 lanes for 128 */
  int incl_all = njobs;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(incl_all, (unsigned)d, 64); if (lane >= d) incl_all += t; }
  const int total = __shfl(incl_all, 63, 64);
  const int excl_all = incl_all - njobs;
  uint32_t dst0 = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) dst0 = cls == k ? base[k] + (uint32_t)(incl[k] - njobs) : dst0;
  for (int t0 = 0; t0 < total; t0 += 64) {
    const int t = t0 + lane;
    int k = 0;
#pragma unroll
    for (int step = 32; step >= 1; step >>= 1) {
      const int cand = k + step;
      const int e = __shfl(excl_all, cand & 63, 64);
      if (cand < 64 && e <= t) k = cand;
    }
    const int local = t - __shfl(excl_all, k, 64);
    const int ns_k = __shfl(ns, k, 64);
    const uint32_t dst = __shfl(dst0, k, 64) + (uint32_t)local;
    const uint32_t pbi = (uint32_t)__shfl(i, k, 64);
    if (t < total && dst < p.jobs_cap) {
      const int r = local / ns_k, s = local - r * ns_k;
      p.jobs[dst] = pbi | ((uint32_t)s << 25) | ((uint32_t)r << 29);
    }
  }
  if (!active || !p.fill_pb_of_in_meta) return;
  for (int y = pb.y >> 2; y < ((pb.y + pb.h) >> 2) && y < p.h4; y++)
    for (int x = pb.x >> 2; x < ((pb.x + pb.w) >> 2) && x < p.w4; x++) p.pb_of[y * p.w4 + x] = (uint32_t)i + 1;
}

__global__ void __launch_bounds__(256) k_meta_pb(DevPic p)
{
  M355_GATE(p);
  k_meta_pb_body(p, (int)blockIdx.x, (int)gridDim.x);
}

/* one thread per (CTB, component): which of the 3x3 neighbouring CTBs may NOT contribute SAO edge
 * neighbours (sao.cc:122-164): outside the picture, another slice with filtering across it disabled, or
 * another tile with loop_filter_across_tiles off.  Reproduces the reference's quirk of looking up the
 * CTB's own slice address with COMPONENT coordinates used as luma coordinates (sao.cc:56), which is
 * why the centre CTB has a bit too.  k_sao then needs no dependent global loads per border sample. */
__device__ __forceinline__ void k_meta_sao_body(const DevPic& p, const int blk)
{
  M355_GATE(p);
  const int i = blk * blockDim.x + threadIdx.x;
  const int nc = p.pp.chroma_format_idc ? 3 : 1;
  if (i >= p.nCtb * nc) return;
  const int c = i / p.nCtb, ctb = i - c * p.nCtb;
  const int xCtb = ctb % p.ctbW, yCtb = ctb / p.ctbW;
  const int csw = c ? (p.sw == 2) : 0, csh = c ? (p.sh == 2) : 0;
  const int l2w = p.pp.log2_ctb_size - csw, l2h = p.pp.log2_ctb_size - csh;
  const int xC = xCtb << l2w, yC = yCtb << l2h;
  const int ctbSliceAddrRS = d_slice_at(p, min(xC, p.pp.width - 1), min(yC, p.pp.height - 1)).slice_addr_rs;
  const int curFlags = p.slices[p.ctbs[ctb].slice_idx].flags;
  const bool across_tiles = (p.pp.flags & M355_PF_LF_ACROSS_TILES) != 0;
  uint32_t mask = 0;
  for (int k = 0; k < 9; k++) {
    const int nx = xCtb + k % 3 - 1, ny = yCtb + k / 3 - 1;
    bool blocked = true;
    if (nx >= 0 && ny >= 0 && nx < p.ctbW && ny < p.ctbH) {
      const int n = ny * p.ctbW + nx;
      const m355_slice shN = p.slices[p.ctbs[n].slice_idx];
      blocked = (shN.slice_addr_rs < ctbSliceAddrRS && !(curFlags & M355_SF_LF_ACROSS_SLICES)) ||
                (shN.slice_addr_rs > ctbSliceAddrRS && !(shN.flags & M355_SF_LF_ACROSS_SLICES)) ||
                (!across_tiles && p.tile_id[n] != p.tile_id[ctb]);
    }
    if (blocked) mask |= 1u << k;
  }
  /* bit 15: the CTB's slice runs SAO on this component (slice_sao_luma / chroma_flag) — k_sao then needs no slice record at all */
  if (curFlags & (c == 0 ? M355_SF_SAO_LUMA : M355_SF_SAO_CHROMA)) mask |= 0x8000u;
  p.sao_nb[i] = (uint16_t)mask;
}

/* the inter stage needs only the job list (+ pb_of when the inter stage is off) */
void m355_launch_job_count(const DevPic& p, bool clear_planes, hipStream_t st)
{
  if (!p.n_pbs) return;
  hipLaunchKernelGGL(k_job_count, dim3((p.n_pbs + 255) / 256), dim3(256), 0, st, p, clear_planes ? 1 : 0);
}
void m355_launch_job_list(const DevPic& p, hipStream_t st)       /* behind m355_launch_job_count */
{
  if (!p.n_pbs) return;
  const int n_chunks = (p.n_pbs + 255) / 256;
  hipLaunchKernelGGL(k_meta_pb, dim3(n_chunks), dim3(256), 0, st, p);
}
void m355_launch_meta_jobs(const DevPic& p, hipStream_t st)
{
  m355_launch_job_count(p, false, st);
  m355_launch_job_list(p, st);
}

/* the CU plane + PB edges and the SAO neighbour masks as ONE launch (independent scatters: blocks [0, nb_cu) walk the CUs, the rest
   the CTBs); the transform edges follow in a launch of their own — a leaf looks its CU up in the plane the first launch wrote */
__global__ void __launch_bounds__(256) k_meta_planes(DevPic p, int nb_cu)
{
  const int b = (int)blockIdx.x;
  if (b < nb_cu) k_meta_cu_body(p, b);
  else k_meta_sao_body(p, b - nb_cu);
}
/* ... and, on a lane that runs a picture on ONE stream (up to 4K), the job list of k_inter_jobs as a third role of the same launch: the
   three scatters are independent of each other, the job list only needs k_job_count's counts (the launch in front).  One packet less per
   picture, and the two latency-bound scatters run beside each other instead of one after the other. */
__global__ void __launch_bounds__(256) k_meta_planes_jobs(DevPic p, int nb_pb, int nb_cu)
{
  M355_GATE(p);
  const int b = (int)blockIdx.x;
  if (b < nb_pb) k_meta_pb_body(p, b, nb_pb);
  else if (b < nb_pb + nb_cu) k_meta_cu_body(p, b - nb_pb);
  else k_meta_sao_body(p, b - nb_pb - nb_cu);
}
__global__ void __launch_bounds__(256) k_meta_tu(DevPic p) { k_meta_tu_body(p, (int)blockIdx.x); }
__global__ void __launch_bounds__(256) k_meta_cu_batch(DevBatch b) { M355_BATCH_PIC(b); k_meta_cu_body(p, (int)blockIdx.x); }
__global__ void __launch_bounds__(256) k_meta_tu_batch(DevBatch b) { M355_BATCH_PIC(b); k_meta_tu_body(p, (int)blockIdx.x); }
__global__ void __launch_bounds__(256) k_meta_sao_batch(DevBatch b) { M355_BATCH_PIC(b); k_meta_sao_body(p, (int)blockIdx.x); }
/* the zero fill in front of them (one plane of the grid per picture; the region is 16-byte aligned and padded, as below) */
__global__ void __launch_bounds__(256) k_meta_fill_batch(DevBatch b)
{
  M355_BATCH_PIC(b);
  const size_t n16 = d_meta_fill16(p);
  uint4* q = (uint4*)p.edge_tu;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) q[i] = make_uint4(0, 0, 0, 0);
}

/* metadata planes for intra availability, deblocking and SAO (not read by k_inter / k_residual) */
void m355_launch_meta_planes(const DevPic& p, hipStream_t st, bool cleared, bool with_tu)
{
  /* edge_tu, edge_pb (sparse writers) and cb_cu (robustness against uncovered areas) live in ONE allocation: one
     fill (`cleared`: k_job_count did it).  pb_of needs none: it is only read where both sides are inter-coded, i.e. covered by a PB. */
  if (!cleared) hipMemsetAsync(p.edge_tu, 0, (size_t)p.w4 * p.h4 * 2 + (size_t)p.wcb * p.hcb * 4 + 64, st);
  const int nb_cu = (p.n_cus + 255) / 256, nb_sao = (p.pp.flags & M355_PF_SAO_ENABLED) ? (p.nCtb * 3 + 255) / 256 : 0;
  if (nb_cu + nb_sao) hipLaunchKernelGGL(k_meta_planes, dim3(nb_cu + nb_sao), dim3(256), 0, st, p, nb_cu);
  if (with_tu && p.n_tus) hipLaunchKernelGGL(k_meta_tu, dim3((p.n_tus + 255) / 256), dim3(256), 0, st, p);   /* (else: k_tu_plan, k_intra.hip) */
}

/* metadata planes (without the transform edges: m355_launch_tu_plan follows) + the job list in one launch; the planes were cleared and the
   jobs counted by k_job_count in front */
void m355_launch_meta_planes_jobs(const DevPic& p, hipStream_t st)
{
  const int nb_pb = (p.n_pbs + 255) / 256, nb_cu = (p.n_cus + 255) / 256, nb_sao = (p.pp.flags & M355_PF_SAO_ENABLED) ? (p.nCtb * 3 + 255) / 256 : 0;
  if (nb_pb + nb_cu + nb_sao) hipLaunchKernelGGL(k_meta_planes_jobs, dim3(nb_pb + nb_cu + nb_sao), dim3(256), 0, st, p, nb_pb, nb_cu);
}

void m355_launch_meta_planes_batch(const HostBatch& b, hipStream_t st)
{
  int n_cus = 0, n_tus = 0, n_sao = 0; size_t fill = 0; uint32_t sao_on = 0;
  for (int k = 0; k < b.n; k++) {
    if (!((b.on >> k) & 1u)) continue;
    const DevPic& p = b.host[k];
    n_cus = std::max(n_cus, p.n_cus); n_tus = std::max(n_tus, p.n_tus);
    fill = std::max(fill, d_meta_fill16(p));
    if (p.pp.flags & M355_PF_SAO_ENABLED) { sao_on |= 1u << k; n_sao = std::max(n_sao, p.nCtb * 3); }
  }
  const DevBatch d{b.dev, b.on};
  if (fill) hipLaunchKernelGGL(k_meta_fill_batch, dim3((unsigned)std::min<size_t>((fill + 255) / 256, 1024), 1, b.n), dim3(256), 0, st, d);
  if (n_cus) hipLaunchKernelGGL(k_meta_cu_batch, dim3((n_cus + 255) / 256, 1, b.n), dim3(256), 0, st, d);
  if (n_tus) hipLaunchKernelGGL(k_meta_tu_batch, dim3((n_tus + 255) / 256, 1, b.n), dim3(256), 0, st, d);
  if (n_sao) hipLaunchKernelGGL(k_meta_sao_batch, dim3((n_sao + 255) / 256, 1, b.n), dim3(256), 0, st, DevBatch{b.dev, sao_on});
}

/* zero fill that a rejected decode (k_validate) does not perform: planes are 128-byte-pitched allocations with a 256-byte tail */
__global__ void __launch_bounds__(256) k_clear_gated(DevPic p, uint4* q, size_t n16)
{
  M355_GATE(p);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) q[i] = make_uint4(0, 0, 0, 0);
}
void m355_launch_clear_gated(const DevPic& p, void* ptr, size_t bytes, hipStream_t st)
{
  const size_t n16 = (bytes + 15) / 16;
  if (!n16) return;
  const unsigned grid = (unsigned)std::min<size_t>((n16 + 255) / 256, 4096);
  hipLaunchKernelGGL(k_clear_gated, dim3(grid), dim3(256), 0, st, p, (uint4*)ptr, n16);
}

void m355_launch_meta(const DevPic& p, hipStream_t st)
{
  m355_launch_meta_planes(p, st, false, true);
  m355_launch_meta_jobs(p, st);
}


/* ---- device-side validation of work lists that were recorded in place (m355_arena_begin): the record checks of the host's
 * validate() (runtime_upload.hip), one thread per record over the concatenation cus | tus | pbs | wts | rbs (4 bins) | ibs.  A rejected
 * record writes the decode's epoch into the lane's gate word (every later kernel of THIS decode returns at once: bad lists are
 * never acted upon; nothing needs resetting for the next decode) and leaves (list << 28 | record), tagged with the epoch, for the
 * per-decode status the host reads back (m355_decode_status, m355_wait).  (The CTB table and each CTB's intra block geometry
 * are checked on the host: its schedules index by them.) ---- */
__global__ void __launch_bounds__(256) k_validate(DevPic p, uint32_t n_total)
{
  const uint32_t g = blockIdx.x * 256u + threadIdx.x;
  if (g >= n_total) return;
  const DevPicParams& pp = p.pp;
  const uint32_t cnt[9] = {(uint32_t)p.n_cus, (uint32_t)p.n_tus, (uint32_t)p.n_pbs, (uint32_t)p.n_wts, (uint32_t)p.rb_count[0], (uint32_t)p.rb_count[1],
                           (uint32_t)p.rb_count[2], (uint32_t)p.rb_count[3], (uint32_t)p.n_ibs};
  uint32_t i = g;
  int q = 0;
  while (q < 8 && i >= cnt[q]) { i -= cnt[q]; q++; }
  bool bad = false;
  const int W = pp.width, H = pp.height;
  if (q == 0) {
    const m355_cu cu = p.cus[i];
    bad = cu.log2_size < pp.log2_min_cb_size || cu.log2_size > pp.log2_ctb_size || cu.x >= W || cu.y >= H || cu.pred_mode > 2 || cu.part_mode > 7;
  } else if (q == 1) {
    const m355_tu tu = p.tus[i];
    bad = tu.log2_size < 2 || tu.log2_size > 6 || tu.x >= W || tu.y >= H;
  } else if (q == 2) {
    const m355_pb pb = p.pbs[i];
    bad = pb.w < 4 || pb.h < 4 || pb.w > 64 || pb.h > 64 || (pb.w & 3) || (pb.h & 3) || pb.x + pb.w > W || pb.y + pb.h > H || !(pb.flags & (M355_PBF_MC_L0 | M355_PBF_MC_L1));
    for (int l = 0; l < 2 && !bad; l++) {
      if (!(pb.flags & (M355_PBF_MC_L0 << l))) continue;
      if (!(pb.flags & (M355_PBF_FILL_L0 << l)) && (pb.ref_slot[l] < 0 || pb.ref_slot[l] >= M355_MAX_REF_FRAMES || !((p.ref_valid >> pb.ref_slot[l]) & 1u))) bad = true;
      if ((pb.flags & M355_PBF_WEIGHTED) && pb.wt_idx[l] >= p.n_wts) bad = true;
    }
  } else if (q == 3) {
    const m355_wt wt = p.wts[i];
    bad = wt.log2wd_luma < 1 || wt.log2wd_luma > 31 || (pp.chroma_format_idc && (wt.log2wd_chroma < 1 || wt.log2wd_chroma > 31));
  } else if (q <= 7) {
    const int sb = q - 4;
    const m355_rb rb = p.rb_bin[sb][i];
    const int n = 1 << (sb + 2);
    const int Wc = rb.cidx ? W / p.sw : W, Hc = rb.cidx ? H / p.sh : H;
    bad = rb.log2_size != sb + 2 || rb.cidx > 2 || rb.kind > 3 || rb.x + n > Wc || rb.y + n > Hc ||
          (unsigned long long)rb.coeff_ofs + rb.ncoeff > p.n_coeffs ||
          ((rb.flags & M355_RBF_DEFERRED) && (unsigned long long)rb.res_ofs + (unsigned)(n * n) > p.res_len) ||
          ((pp.flags & M355_PF_SCALING_LIST) && (rb.matrix_id & 7) > 5) || (rb.kind == M355_RK_DST && sb != 0);
  } else {
    const m355_ib ib = p.ibs[i];          /* (the sorted copy: same records as the caller's) */
    const int n = 1 << (ib.log2_size & 7);
    const int Wc = ib.cidx ? W / p.sw : W, Hc = ib.cidx ? H / p.sh : H;
    bad = ib.log2_size < 2 || ib.log2_size > 5 || ib.cidx > 2 || ib.mode > 34 || ib.x + n > Wc || ib.y + n > Hc ||
          ((ib.flags & M355_IBF_HAS_RESIDUAL) && (unsigned long long)ib.res_ofs + (unsigned)(n * n) > p.res_len) ||
          ((ib.flags & M355_IBF_PCM) && (unsigned long long)ib.res_ofs + (unsigned)(n * n) > p.n_pcm);
  }
  if (bad) {
    const int list = q < 4 ? q + 1 : (q < 8 ? 5 : 6);       /* numbering of the host's messages: cu tu pb weight rb ib = 1..6 */
    p.timeout[1] = p.epoch;                                  /* this decode's gate (every rejecting thread stores the same value) */
    atomicMin((unsigned long long*)(p.timeout + 2), ((unsigned long long)(~p.epoch) << 32) | (((uint32_t)list << 28) | (i < 0x0FFFFFFFu ? i : 0x0FFFFFFFu)));
  }
}

void m355_launch_validate(const DevPic& p, hipStream_t st)
{
  const uint64_t n = (uint64_t)p.n_cus + p.n_tus + p.n_pbs + p.n_wts + p.rb_count[0] + p.rb_count[1] + p.rb_count[2] + p.rb_count[3] + p.n_ibs;
  if (!n) return;
  hipLaunchKernelGGL(k_validate, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p, (uint32_t)n);
}
