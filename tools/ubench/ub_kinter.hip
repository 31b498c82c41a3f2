// k_inter_jobs<u16> outside the library: the PRODUCT kernel (this file includes libde265_amd/csrc/k_inter.hip) on a synthetic picture whose block mix, vectors and
// list use this harness controls, beside cut-down outer kernels around the same job function (d_inter_job_lean) — where does the time go that the traffic model
// (ub_tile.hip: the same loads and stores, a checksum instead of the filters) does not need?
//   V0  the product kernel as launched by the library (gate word, range ends from device memory, reference table + tap tables to LDS, all three job modes)
//   V1  no gate word, range ends as kernel arguments (two dependent round trips less in front of the job word)
//   V2  V1 with the main job mode only in the kernel (no explicit-weight / EDGE code: 120 registers instead of 158), still 3 waves per SIMD
//   V3  V2 at 4 waves per SIMD
// Every variant's destination planes are compared with V0's.
// usage: ub_kinter [bipred_pct (default 100)]
#include "../../libde265_amd/csrc/k_inter.hip"
#include <stdio.h>
#include <vector>
#include <algorithm>
#include <string.h>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
#define PW 7680
#define PH 4320

template <int V>
__global__ void __launch_bounds__(256, V == 3 ? 4 : 3) k_var(DevPic p, int t0, int t1, int zero)
{
  auto blocks8 = [](int jobs) { return (((jobs + 255) / 256 + 7) / 8) * 8; };
  const int nblk_uni8 = blocks8(t0), nblk_bi8 = blocks8(t1 - t0);
  if ((int)blockIdx.x >= nblk_bi8 + nblk_uni8) return;
  const int b = blockIdx.x, xcd = b & 7, slot = b >> 3;
  const int per_bi = nblk_bi8 >> 3, per_uni = nblk_uni8 >> 3, per = per_bi + per_uni;
  const int bi_before = (int)(((long long)slot * per_bi) / per), bi_after = (int)(((long long)(slot + 1) * per_bi) / per);
  int cls, ji, jend;
  if (bi_after != bi_before) { cls = 1; ji = t0 + (xcd * per_bi + bi_before) * 256; jend = t1; }
  else { cls = 0; ji = (xcd * per_uni + slot - bi_before) * 256; jend = t0; }
  if (ji >= jend) return;
  ji += threadIdx.x;
  __shared__ DevRef s_refs[M355_MAX_REF_FRAMES];
  {
    const unsigned* src = (const unsigned*)p.refs;
    unsigned* dst = (unsigned*)s_refs;
    for (int i = threadIdx.x; i < (int)(sizeof(s_refs) / 4); i += 256) dst[i] = src[i];
  }
  __shared__ __attribute__((aligned(16))) unsigned s_tab[LT_WORDS];
  if (threadIdx.x < LT_WORDS / 4) ((uint4*)s_tab)[threadIdx.x] = ((const uint4*)p.inter_tabs)[threadIdx.x];
  uint32_t job = 0;
  if (ji < jend) job = p.jobs[ji];
  __syncthreads();
  if (ji >= jend) return;
  __shared__ __attribute__((aligned(16))) unsigned s_ext[256 * 20];
  unsigned* ext = s_ext + threadIdx.x * 20;
  if (V == 1) {
    cls += zero;                                   /* (opaque to the compiler: the other modes' code stays) */
    /* (the other modes stay in the kernel: never taken here, but they set its register count) */
    if (cls == 3) d_inter_job_lean<uint16_t, 2>(p, job, false, s_tab + LT_QL, s_tab + LT_QV, s_tab + LT_CL, s_tab + LT_CV, ext, s_refs);
    else if (cls == 2) d_inter_job_lean<uint16_t, 1>(p, job, false, s_tab + LT_QL, s_tab + LT_QV, s_tab + LT_CL, s_tab + LT_CV, ext, s_refs);
    else d_inter_job_lean<uint16_t, 0>(p, job, cls == 1, s_tab + LT_QL, s_tab + LT_QV, s_tab + LT_CL, s_tab + LT_CV, ext, s_refs);
  } else d_inter_job_lean<uint16_t, 0>(p, job, cls == 1, s_tab + LT_QL, s_tab + LT_QV, s_tab + LT_CL, s_tab + LT_CV, ext, s_refs);
}

/* a streaming read of n 16-byte words (the prefetch: what it leaves in the L2s / the Infinity Cache is the point) and a streaming write (the flush) */
__global__ void __launch_bounds__(256) k_stream_read(const uint4* src, size_t n, unsigned* sink)
{
  unsigned acc = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { const uint4 v = src[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
  if (acc == 0x12345678u) sink[0] = acc;
}
__global__ void __launch_bounds__(256) k_stream_write(uint4* dst, size_t n, unsigned v)
{
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = make_uint4(v, v, v, v);
}
typedef unsigned v4u __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) k_stream_write_nt(v4u* dst, size_t n, unsigned v)
{
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) __builtin_nontemporal_store(v4u{v, v, v, v}, dst + i);
}
__global__ void __launch_bounds__(256) k_stream_read_nt(const v4u* src, size_t n, unsigned* sink)
{
  unsigned acc = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { const v4u v = __builtin_nontemporal_load(src + i); acc ^= v.x ^ v.y ^ v.z ^ v.w; }
  if (acc == 0x12345678u) sink[0] = acc;
}
static unsigned rnd_state = 12345;
static unsigned rnd() { rnd_state ^= rnd_state << 13; rnd_state ^= rnd_state >> 17; rnd_state ^= rnd_state << 5; return rnd_state; }
static int bipred_pct = 100;

static void gen_pbs(std::vector<m355_pb>& pbs, int x, int y, int size, int S)
{
  if (size > S) { const int h = size / 2; for (int i = 0; i < 4; i++) gen_pbs(pbs, x + (i & 1) * h, y + (i >> 1) * h, h, S); return; }
  if (x >= PW || y >= PH) return;
  m355_pb pb; memset(&pb, 0, sizeof pb);
  pb.x = (uint16_t)x; pb.y = (uint16_t)y; pb.w = (uint8_t)S; pb.h = (uint8_t)std::min(S, PH - y);
  const bool bi = (int)(rnd() % 100) < bipred_pct;
  const int only = rnd() & 1;
  pb.ref_slot[0] = pb.ref_slot[1] = -1;
  for (int l = 0; l < 2; l++) {
    if (!bi && l != only) continue;
    pb.flags |= (uint8_t)((M355_PBF_PRED_L0 | M355_PBF_MC_L0) << l);
    pb.ref_slot[l] = (int8_t)(rnd() & 1);
    int mvx = (int)(rnd() % 129) - 64, mvy = (int)(rnd() % 129) - 64;
    /* the whole PB's windows inside the picture (the EDGE class is not what is measured here) */
    mvx = std::min(std::max(mvx, (32 - x) * 4), (PW - 40 - S - x) * 4); mvy = std::min(std::max(mvy, (32 - y) * 4), (PH - 40 - S - y) * 4);
    pb.mv[l][0] = (int16_t)mvx; pb.mv[l][1] = (int16_t)mvy;
  }
  pbs.push_back(pb);
}

int main(int argc, char** argv)
{
  if (argc > 1) bipred_pct = atoi(argv[1]);
  // reference frames (two, three planes each) and the destination
  DevRef href[M355_MAX_REF_FRAMES]; memset(href, 0, sizeof href);
  std::vector<uint16_t> h((size_t)PW * PH);
  for (int f = 0; f < 2; f++)
    for (int c = 0; c < 3; c++) {
      const int pw = c ? PW / 2 : PW, ph = c ? PH / 2 : PH;
      for (size_t i = 0; i < (size_t)pw * ph; i++) h[i] = (uint16_t)(((i * 2654435761u) ^ (i >> 11) ^ (77u * f + 13u * c)) >> 5 & 0x3FF);
      void* d; CHK(hipMalloc(&d, (size_t)pw * ph * 2 + 256)); CHK(hipMemcpy(d, h.data(), (size_t)pw * ph * 2, hipMemcpyHostToDevice));
      href[f].plane[c] = d; href[f].stride[c] = pw; href[f].valid = 1;
    }
  DevPic p; memset(&p, 0, sizeof p);
  p.pp.width = PW; p.pp.height = PH; p.pp.chroma_format_idc = 1; p.pp.bit_depth_luma = 10; p.pp.bit_depth_chroma = 10; p.pp.log2_ctb_size = 6;
  p.sw = p.sh = 2; p.w4 = PW / 4; p.h4 = PH / 4;
  void* dst[2][3];
  for (int c = 0; c < 3; c++) {
    p.pw[c] = c ? PW / 2 : PW; p.ph[c] = c ? PH / 2 : PH; p.stride[c] = p.pw[c];
    for (int k = 0; k < 2; k++) CHK(hipMalloc(&dst[k][c], (size_t)p.pw[c] * p.ph[c] * 2 + 256));
  }
  { DevRef* d; CHK(hipMalloc(&d, sizeof href)); CHK(hipMemcpy(d, href, sizeof href, hipMemcpyHostToDevice)); p.refs = d; }
  { uint32_t tabs[M355_INTER_TAB_WORDS + 4]; m355_inter_tables(false, 10, 10, tabs); uint32_t* d; CHK(hipMalloc(&d, sizeof tabs)); CHK(hipMemcpy(d, tabs, sizeof tabs, hipMemcpyHostToDevice)); p.inter_tabs = d; }
  { uint32_t* d; CHK(hipMalloc(&d, (size_t)p.w4 * p.h4 * 4)); p.pb_of = d; }
  { uint32_t* d; CHK(hipMalloc(&d, 64)); CHK(hipMemset(d, 0, 64)); p.timeout = d; p.epoch = 7; }
  hipEvent_t e0, e1, e2; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1)); CHK(hipEventCreate(&e2));
  const size_t flush_bytes = (size_t)1 << 30;
  uint4* flush; CHK(hipMalloc(&flush, flush_bytes));
  unsigned* sink; CHK(hipMalloc(&sink, 64));

  const int mixes[][4] = {{8, 16, 32, 64}, {64, 64, 64, 64}, {16, 16, 16, 16}, {8, 8, 8, 8}};
  const char* mixname[] = {"mix 8/16/32/64 per CTB", "64x64 only", "16x16 only", "8x8 only"};
  printf("bipred_pct %d\n", bipred_pct);
  for (int m = 0; m < 4; m++) {
    std::vector<m355_pb> pbs;
    rnd_state = 12345;
    for (int cy = 0; cy < PH; cy += 64) for (int cx = 0; cx < PW; cx += 64) gen_pbs(pbs, cx, cy, 64, mixes[m][rnd() & 3]);
    // job list as k_meta_pb leaves it: one-list jobs, then two-list jobs, each in PB order, strips left to right, row blocks top to bottom
    std::vector<uint32_t> jobs[2];
    for (size_t i = 0; i < pbs.size(); i++) {
      const m355_pb& pb = pbs[i];
      const int bi = (pb.flags & M355_PBF_MC_L0) && (pb.flags & M355_PBF_MC_L1);
      for (int rb = 0; rb < (pb.h + 7) / 8; rb++) for (int st = 0; st < pb.w / 4; st++) jobs[bi].push_back((uint32_t)i | ((uint32_t)st << 25) | ((uint32_t)rb << 29));
    }
    const int t0 = (int)jobs[0].size(), t1 = t0 + (int)jobs[1].size();
    std::vector<uint32_t> all(jobs[0]); all.insert(all.end(), jobs[1].begin(), jobs[1].end());
    { m355_pb* d; CHK(hipMalloc(&d, pbs.size() * sizeof(m355_pb))); CHK(hipMemcpy(d, pbs.data(), pbs.size() * sizeof(m355_pb), hipMemcpyHostToDevice)); p.pbs = d; p.n_pbs = (int)pbs.size(); }
    { uint32_t* d; CHK(hipMalloc(&d, all.size() * 4 + 64)); CHK(hipMemcpy(d, all.data(), all.size() * 4, hipMemcpyHostToDevice)); p.jobs = d; p.jobs_cap = (uint32_t)all.size(); }
    { uint32_t tot[4] = {(uint32_t)t0, (uint32_t)t1, (uint32_t)t1, (uint32_t)t1}; uint32_t* d; CHK(hipMalloc(&d, 16)); CHK(hipMemcpy(d, tot, 16, hipMemcpyHostToDevice)); p.job_tot = d; }
    printf("%-26s %7zu PBs, %8d jobs (%d from one list)\n", mixname[m], pbs.size(), t1, t0);
    const unsigned grid = (unsigned)((p.jobs_cap + 255) / 256) + 4 * 8;
    std::vector<uint16_t> ref0, got;
    for (int v = 0; v < 4; v++) {
      for (int c = 0; c < 3; c++) { p.plane[c] = dst[v ? 1 : 0][c]; CHK(hipMemset(p.plane[c], 0, (size_t)p.pw[c] * p.ph[c] * 2)); }
      std::vector<float> t;
      for (int r = 0; r < 8; r++) {
        CHK(hipEventRecord(e0));
        if (v == 0) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_inter_jobs<uint16_t>), dim3(grid), dim3(256), 0, 0, p);
        else if (v == 1) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_var<1>), dim3(grid), dim3(256), 0, 0, p, t0, t1, 0);
        else if (v == 2) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_var<2>), dim3(grid), dim3(256), 0, 0, p, t0, t1, 0);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_var<3>), dim3(grid), dim3(256), 0, 0, p, t0, t1, 0);
        CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
        if (r) t.push_back(ms);
      }
      std::sort(t.begin(), t.end());
      // compare with V0
      bool same = true;
      if (v == 0) { ref0.resize((size_t)PW * PH * 3 / 2); size_t o = 0; for (int c = 0; c < 3; c++) { CHK(hipMemcpy(ref0.data() + o, dst[0][c], (size_t)p.pw[c] * p.ph[c] * 2, hipMemcpyDeviceToHost)); o += (size_t)p.pw[c] * p.ph[c]; } }
      else { got.resize(ref0.size()); size_t o = 0; for (int c = 0; c < 3; c++) { CHK(hipMemcpy(got.data() + o, dst[1][c], (size_t)p.pw[c] * p.ph[c] * 2, hipMemcpyDeviceToHost)); o += (size_t)p.pw[c] * p.ph[c]; } same = memcmp(got.data(), ref0.data(), ref0.size() * 2) == 0; }
      const char* what[] = {"V0 product kernel", "V1 no gate, ends in arguments", "V2 main mode only, 3 waves", "V3 main mode only, 4 waves"};
      printf("    %-32s %.4f ms (min %.4f)%s\n", what[v], t[t.size() / 2], t[0], v ? (same ? "  = V0" : "  DIFFERS from V0") : "");
    }
    /* the product kernel by the state of the caches: as above (launch after launch: the two reference frames, 199 MB, largely stay in the 256 MB Infinity Cache),
       behind a 1 GiB streaming write (cold), and behind that write + a streaming read of both reference frames (prefetched) */
    for (int c = 0; c < 3; c++) p.plane[c] = dst[0][c];
    for (int mode = 0; mode < 7; mode++) {
      std::vector<float> t, tp;
      for (int r = 0; r < 6; r++) {
        const size_t n16 = flush_bytes / 16;
        if (mode == 1 || mode == 2) hipLaunchKernelGGL(k_stream_write, dim3(256 * 8), dim3(256), 0, 0, flush, n16, (unsigned)r);
        if (mode == 3) hipLaunchKernelGGL(k_stream_write_nt, dim3(256 * 8), dim3(256), 0, 0, (v4u*)flush, n16, (unsigned)r);
        if (mode == 4) hipLaunchKernelGGL(k_stream_read, dim3(256 * 8), dim3(256), 0, 0, (const uint4*)flush, n16, sink);
        if (mode == 5) hipLaunchKernelGGL(k_stream_read_nt, dim3(256 * 8), dim3(256), 0, 0, (const v4u*)flush, n16, sink);
        if (mode == 6) { hipLaunchKernelGGL(k_stream_read_nt, dim3(256 * 8), dim3(256), 0, 0, (const v4u*)flush, n16 / 2, sink); hipLaunchKernelGGL(k_stream_write_nt, dim3(256 * 8), dim3(256), 0, 0, (v4u*)flush + n16 / 2, n16 / 2, (unsigned)r); }
        CHK(hipEventRecord(e2));
        if (mode == 2)
          for (int f = 0; f < 2; f++)
            for (int c = 0; c < 3; c++) hipLaunchKernelGGL(k_stream_read, dim3(256 * 8), dim3(256), 0, 0, (const uint4*)href[f].plane[c], (size_t)p.pw[c] * p.ph[c] * 2 / 16, sink);
        CHK(hipEventRecord(e0));
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_inter_jobs<uint16_t>), dim3(grid), dim3(256), 0, 0, p);
        CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
        float ms, msp; CHK(hipEventElapsedTime(&ms, e0, e1)); CHK(hipEventElapsedTime(&msp, e2, e0));
        if (r) { t.push_back(ms); tp.push_back(msp); }
      }
      std::sort(t.begin(), t.end()); std::sort(tp.begin(), tp.end());
      const char* st[] = {"launch after launch (warm)", "behind a 1 GiB streaming write (cold)", "cold + both reference frames streamed in first", "behind a 1 GiB NON-TEMPORAL streaming write",
                          "behind a 1 GiB streaming read", "behind a 1 GiB NON-TEMPORAL streaming read", "behind 512 MiB nt read + 512 MiB nt write"};
      printf("    V0 %-48s %.4f ms\n", st[mode], t[t.size() / 2]);
      if (mode == 2) printf("       (the streaming read of the 199 MB itself: %.4f ms)\n", tp[tp.size() / 2]);
    }
    CHK(hipFree((void*)p.pbs)); CHK(hipFree(p.jobs)); CHK(hipFree(p.job_tot));
  }
  return 0;
}
