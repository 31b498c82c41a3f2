// k_inter_jobs<u16>'s traffic model (ub_tile.hip, linear planes) with the product kernel's STRUCTURE as knobs: how many window rows are requested together and how far
// ahead of the arithmetic (the product: row pairs, two pairs ahead), how much vector arithmetic stands between two requests (the product: ~40 dot2 + packs per row
// pair; ~180 more behind a list's last row, with nothing of the wave in flight), and the order of the phases (the product: luma L0, luma L1, luma stores, chroma L0,
// chroma L1, chroma stores).  What turns the model's 0.08 ms into the product's 0.16?
// usage: ub_inter_model
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef unsigned u4 __attribute__((ext_vector_type(4)));
typedef unsigned u3 __attribute__((ext_vector_type(3)));
typedef unsigned u2 __attribute__((ext_vector_type(2)));
typedef short s2 __attribute__((ext_vector_type(2)));
#define G __attribute__((address_space(1)))
#define W 7680
#define H 4320

struct Job { unsigned short x0, y0; signed char mv[2][2]; };
struct Planes { const unsigned char* ref[2][3]; unsigned char* dst[3]; };
#define PIN(x) asm volatile("" : "+v"(x) : : "memory")
__device__ __forceinline__ int dot2(unsigned a, unsigned b, int c) { return __builtin_amdgcn_sdot2(__builtin_bit_cast(s2, a), __builtin_bit_cast(s2, b), c, false); }

__device__ __forceinline__ void load_luma(const G unsigned char* plane, int xa, int y, unsigned* r)
{
  const G unsigned char* q = plane + (size_t)y * (W * 2) + (size_t)(xa & ~1) * 2;
  const u4 a = *(const G u4*)q; const u2 b = *(const G u2*)(q + 16);
  r[0] = a.x; r[1] = a.y; r[2] = a.z; r[3] = a.w; r[4] = b.x; r[5] = b.y;
}
__device__ __forceinline__ void load_chroma(const G unsigned char* plane, int xa, int y, unsigned* r)
{
  const G unsigned char* q = plane + (size_t)y * W + (size_t)(xa & ~1) * 2;
  const u3 a = *(const G u3*)q;
  r[0] = a.x; r[1] = a.y; r[2] = a.z;
}
/* WORK dot2 per row in four independent chains over the row's registers (the product: 20 per luma row, 6 per chroma row) */
template <int WORK, int NR>
__device__ __forceinline__ void row_work(const unsigned* r, const unsigned* taps, int* acc)
{
  if (WORK == 0) {                                   /* (the loads must not be dead code) */
#pragma unroll
    for (int i = 0; i < NR; i++) acc[i & 3] ^= (int)r[i];
  }
#pragma unroll
  for (int i = 0; i < WORK; i++) acc[i & 3] = dot2(r[i % NR], taps[i % 5], acc[i & 3]);
}

/* GROUP rows requested together, AHEAD groups ahead; WORK dot2 per luma row; VWORK dot2 behind a list's last row; ORDER 0: luma + chroma per list, stores at the end,
   1: the product's (luma L0, luma L1, luma stores, chroma L0, chroma L1, chroma stores) */
template <int GROUP, int AHEAD, int WORK, int VWORK, int ORDER>
__global__ void __launch_bounds__(256, 3) k_model(Planes P, const Job* jobs, int njobs, unsigned* sums, unsigned tapseed)
{
  extern __shared__ unsigned s_pad[];
  if (tapseed == 0xFFFFFFFFu) s_pad[threadIdx.x] = 0;
  const int nb = gridDim.x, b = blockIdx.x, xcd = b & 7, slot = b >> 3, blk = xcd * (nb >> 3) + slot;
  const int ji = blk * 256 + threadIdx.x;
  if (ji >= njobs) return;
  const Job j = jobs[ji];
  unsigned taps[5];
#pragma unroll
  for (int i = 0; i < 5; i++) taps[i] = tapseed * (i + 3);
  int acc[4] = {0, 0, 0, 0}, accc[4] = {0, 0, 0, 0};
  auto luma = [&](int l) {
    const int xa = j.x0 + j.mv[l][0] - 3, ya = j.y0 + j.mv[l][1] - 3;
    const G unsigned char* pl = (const G unsigned char*)P.ref[l][0];
    constexpr int NG = (15 + GROUP - 1) / GROUP, NB = AHEAD + 1;
    unsigned R[NB][GROUP][6];
#pragma unroll
    for (int g = 0; g < AHEAD && g < NG; g++)
#pragma unroll
      for (int r = 0; r < GROUP; r++) if (GROUP * g + r < 15) load_luma(pl, xa, ya + GROUP * g + r, R[g % NB][r]);
#pragma unroll
    for (int g = 0; g < NG; g++) {
      if (g + AHEAD < NG) {
#pragma unroll
        for (int r = 0; r < GROUP; r++) if (GROUP * (g + AHEAD) + r < 15) load_luma(pl, xa, ya + GROUP * (g + AHEAD) + r, R[(g + AHEAD) % NB][r]);
      }
#pragma unroll
      for (int r = 0; r < GROUP; r++) if (GROUP * g + r < 15) row_work<WORK, 6>(R[g % NB][r], taps, acc);
      PIN(acc[0]); PIN(acc[1]); PIN(acc[2]); PIN(acc[3]);
    }
#pragma unroll
    for (int i = 0; i < VWORK; i++) acc[i & 3] = dot2((unsigned)acc[(i + 1) & 3], taps[i % 5], acc[i & 3]);
  };
  auto chroma = [&](int l) {
    const int xc = (j.x0 >> 1) + (j.mv[l][0] >> 1) - 1, yc = (j.y0 >> 1) + (j.mv[l][1] >> 1) - 1;
    unsigned C[2][7][3];
#pragma unroll
    for (int c = 0; c < 2; c++)
#pragma unroll
      for (int r = 0; r < 7; r++) load_chroma((const G unsigned char*)P.ref[l][1 + c], xc, yc + r, C[c][r]);
#pragma unroll
    for (int c = 0; c < 2; c++)
#pragma unroll
      for (int r = 0; r < 7; r++) row_work<(WORK * 3 + 9) / 10, 3>(C[c][r], taps, accc);
#pragma unroll
    for (int i = 0; i < VWORK / 4; i++) accc[i & 3] = dot2((unsigned)accc[(i + 1) & 3], taps[i % 5], accc[i & 3]);
    PIN(accc[0]);
  };
  auto store_luma = [&]() {
    unsigned char* d = P.dst[0] + (size_t)j.y0 * (W * 2) + (size_t)j.x0 * 2;
#pragma unroll
    for (int r = 0; r < 8; r++) __builtin_nontemporal_store(u2{(unsigned)acc[r & 3] + r, (unsigned)acc[(r + 1) & 3]}, (u2*)(d + (size_t)r * (W * 2)));
  };
  auto store_chroma = [&]() {
#pragma unroll
    for (int c = 0; c < 2; c++) {
      unsigned char* dc = P.dst[1 + c] + (size_t)(j.y0 >> 1) * W + (size_t)(j.x0 >> 1) * 2;
#pragma unroll
      for (int r = 0; r < 4; r++) __builtin_nontemporal_store((unsigned)accc[r & 3] + r + c, (unsigned*)(dc + (size_t)r * W));
    }
  };
  if (ORDER == 0) {
#pragma unroll 1
    for (int l = 0; l < 2; l++) { luma(l); chroma(l); }
    store_luma(); store_chroma();
  } else {
#pragma unroll 1
    for (int l = 0; l < 2; l++) luma(l);
    asm volatile("" ::: "memory");
    store_luma();
#pragma unroll 1
    for (int l = 0; l < 2; l++) chroma(l);
    asm volatile("" ::: "memory");
    store_chroma();
  }
  sums[ji] = (unsigned)(acc[0] ^ acc[1] ^ acc[2] ^ acc[3] ^ accc[0] ^ accc[1] ^ accc[2] ^ accc[3]);
}

static unsigned rnd_state = 12345;
static unsigned rnd() { rnd_state ^= rnd_state << 13; rnd_state ^= rnd_state >> 17; rnd_state ^= rnd_state << 5; return rnd_state; }

// jobs of a picture whose CTBs are cut into S x S PBs, S drawn per CTB from `sizes` (the bench workload draws 8 / 16 / 32 / 64 uniformly); PB order = CTB
// raster, z-order inside; every PB: random integer vectors of +-16 per list (synth.c: +-64 quarter samples); strips left to right, row blocks top to bottom
static void gen_pbs(std::vector<Job>& jobs, int x, int y, int size, int S)
{
  if (size > S) { const int h = size / 2; for (int i = 0; i < 4; i++) gen_pbs(jobs, x + (i & 1) * h, y + (i >> 1) * h, h, S); return; }
  if (x >= W || y >= H) return;
  Job j; for (int l = 0; l < 2; l++) for (int k = 0; k < 2; k++) j.mv[l][k] = (signed char)((int)(rnd() % 33) - 16);
  for (int rb = 0; rb < S / 8; rb++)
    for (int st = 0; st < S / 4; st++) {
      Job q = j; q.x0 = (unsigned short)(x + 4 * st); q.y0 = (unsigned short)(y + 8 * rb);
      if (q.y0 + 8 > H) continue;
      for (int l = 0; l < 2; l++) {   // windows inside the picture (the EDGE class is another kernel path)
        q.mv[l][0] = (signed char)(std::min(std::max(q.x0 + q.mv[l][0], 24), W - 48) - q.x0);
        q.mv[l][1] = (signed char)(std::min(std::max(q.y0 + q.mv[l][1], 24), H - 48) - q.y0);
      }
      jobs.push_back(q);
    }
}


template <int GROUP, int AHEAD, int WORK, int VWORK, int ORDER>
static void run(const Planes& P, const Job* dj, int njobs, unsigned* dsums, const char* what)
{
  const int nblk = (((njobs + 255) / 256 + 7) / 8) * 8, lds = (160 * 1024 / 3 - 1024) & ~255;       // 3 workgroups per CU, as the product
  hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  std::vector<float> t;
  for (int r = 0; r < 8; r++) {
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_model<GROUP, AHEAD, WORK, VWORK, ORDER>), dim3(nblk), dim3(256), lds, 0, P, dj, njobs, dsums, 0x00030001u);
    CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    if (r) t.push_back(ms);
  }
  std::sort(t.begin(), t.end());
  printf("    rows per request group %d, groups ahead %d, dot2 per luma row %2d, behind a list %3d, %s: %.4f ms\n", GROUP, AHEAD, WORK, VWORK, ORDER ? "product's phase order" : "list by list         ", t[t.size() / 2]);
  CHK(hipEventDestroy(e0)); CHK(hipEventDestroy(e1));
}

int main()
{
  Planes PL;
  for (int l = 0; l < 2; l++)
    for (int c = 0; c < 3; c++) {
      const int pw = c ? W / 2 : W, ph = c ? H / 2 : H;
      unsigned char* a; CHK(hipMalloc(&a, (size_t)pw * ph * 2 + 4096)); CHK(hipMemset(a, 1 + l + c, (size_t)pw * ph * 2 + 4096));
      PL.ref[l][c] = a;
    }
  for (int c = 0; c < 3; c++) { unsigned char* d; CHK(hipMalloc(&d, (size_t)W * (H + 8) * 2)); PL.dst[c] = d; }
  const int mixes[][4] = {{8, 16, 32, 64}, {64, 64, 64, 64}, {16, 16, 16, 16}};
  const char* mixname[] = {"mix 8/16/32/64 per CTB (the bench workload's), every PB from two lists", "64x64 only", "16x16 only"};
  for (int m = 0; m < 3; m++) {
    std::vector<Job> jobs;
    rnd_state = 12345;
    for (int cy = 0; cy < H; cy += 64) for (int cx = 0; cx < W; cx += 64) gen_pbs(jobs, cx, cy, 64, mixes[m][rnd() & 3]);
    const int njobs = (int)jobs.size();
    Job* dj; CHK(hipMalloc(&dj, (size_t)njobs * sizeof(Job))); CHK(hipMemcpy(dj, jobs.data(), (size_t)njobs * sizeof(Job), hipMemcpyHostToDevice));
    unsigned* s0; CHK(hipMalloc(&s0, (size_t)njobs * 4));
    printf("%s, %d jobs\n", mixname[m], njobs);
    run<3, 2, 0, 0, 0>(PL, dj, njobs, s0, "");      // ub_tile's structure, no arithmetic
    run<2, 2, 0, 0, 0>(PL, dj, njobs, s0, "");      // the product's request structure
    run<2, 2, 0, 0, 1>(PL, dj, njobs, s0, "");
    run<2, 2, 20, 0, 1>(PL, dj, njobs, s0, "");     // + the H pass's arithmetic
    run<2, 2, 20, 180, 1>(PL, dj, njobs, s0, "");   // + the V pass / write-back behind a list (nothing in flight)
    run<2, 3, 20, 180, 1>(PL, dj, njobs, s0, "");
    run<3, 2, 20, 180, 1>(PL, dj, njobs, s0, "");
    run<5, 2, 20, 180, 1>(PL, dj, njobs, s0, "");
    run<15, 1, 20, 180, 1>(PL, dj, njobs, s0, "");  // all rows requested up front
    run<2, 2, 40, 360, 1>(PL, dj, njobs, s0, "");   // twice the arithmetic
    run<2, 2, 10, 90, 1>(PL, dj, njobs, s0, "");    // half
    CHK(hipFree(dj)); CHK(hipFree(s0));
  }
  return 0;
}
