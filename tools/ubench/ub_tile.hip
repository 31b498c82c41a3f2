// Would reference planes stored as LINE-SHAPED tiles (one 128-byte line = 8 x 8 samples of 16 bit) make k_inter_jobs' window fetch faster?
// (MI355X; evidence for DESIGN.md section 4 Round 6, 2c.  Round 4's tiled experiment used 32 x 8-sample tiles with aprons, i.e. 64..96-byte tile ROWS: a
// window row was still a line of its own.  Here a 23 x 23 window touches ~15 lines instead of ~31.)
// Traffic model of k_inter_jobs<u16>'s main class: one LANE per job of 4 luma columns x 8 rows (+ 2 x 4 of Cb and Cr), consecutive lanes = horizontally
// adjacent strips of one PB, workgroup = 256 jobs, XCD-contiguous block order; per list 15 luma window rows of 11 samples and 7 rows of 5 samples per chroma
// plane; 8 + 4 + 4 row stores.  The filter arithmetic is replaced by a checksum of exactly the window's samples, so both layouts can be CHECKED against
// each other (the tiled addressing fetches the same samples) and carry a similar amount of vector arithmetic.
//   layout 0: linear planes (pitch 15360 B luma): per luma row dwordx4 + dwordx2 from the dword-aligned address (as the product), chroma dwordx3
//   layout 1: 8 x 8 line tiles, tile-row major: per luma row the two 16-byte tile rows that hold samples 0..15 from the window's chunk + ONE dword of the
//             third (needed only when the window starts at sample 6 / 7 of its chunk), chroma dwordx4 + dwordx2
// usage: ub_tile [wg_per_cu]
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
#define G __attribute__((address_space(1)))
#define W 7680
#define H 4320

struct Job { unsigned short x0, y0; signed char mv[2][2]; };   // integer part of the motion vectors, luma samples
struct Planes { const unsigned char* ref[2][3]; unsigned char* dst[3]; };

__device__ __forceinline__ unsigned samp(const unsigned* r, int i) { return (r[i >> 1] >> (16 * (i & 1))) & 0xFFFFu; }

// ---- one window row: the raw registers (load_row), then the weighted sum of samples [o, o + N) (sum_row) ----
template <int LAYOUT, int N> struct RowRegs { static constexpr int n = LAYOUT == 0 ? (N == 11 ? 6 : 3) : (N == 11 ? 9 : 6); };
template <int LAYOUT, int N>   // N = 11 (luma) or 5 (chroma); pw = plane width in samples
__device__ __forceinline__ void load_row(const G unsigned char* plane, int pw, int xa, int y, unsigned* r)
{
  if (LAYOUT == 0) {
    const G unsigned char* q = plane + (size_t)y * (pw * 2) + (size_t)(xa & ~1) * 2;
    if (N == 11) {
      const u4 a = *(const G u4*)q; const u2 b = *(const G u2*)(q + 16);
      r[0] = a.x; r[1] = a.y; r[2] = a.z; r[3] = a.w; r[4] = b.x; r[5] = b.y;
    } else {
      const u3 a = *(const G u3*)q;
      r[0] = a.x; r[1] = a.y; r[2] = a.z;
    }
  } else {
    const int c0 = xa >> 3, tpr = pw >> 3;                                  // chunk, tiles per tile row
    const G unsigned char* q = plane + ((size_t)(y >> 3) * tpr + c0) * 128 + (size_t)(y & 7) * 16;
    if (N == 11) {
      const u4 a = *(const G u4*)q; const u4 b = *(const G u4*)(q + 128); r[8] = *(const G unsigned*)(q + 256);
      r[0] = a.x; r[1] = a.y; r[2] = a.z; r[3] = a.w; r[4] = b.x; r[5] = b.y; r[6] = b.z; r[7] = b.w;
    } else {
      const u4 a = *(const G u4*)q; const u2 b = *(const G u2*)(q + 128);
      r[0] = a.x; r[1] = a.y; r[2] = a.z; r[3] = a.w; r[4] = b.x; r[5] = b.y;
    }
  }
}
template <int LAYOUT, int N>
__device__ __forceinline__ unsigned sum_row(const unsigned* r, int xa)
{
  const int o = LAYOUT == 0 ? (xa & 1) : (xa & 7);
  unsigned s = 0;
#pragma unroll
  for (int i = 0; i < 2 * RowRegs<LAYOUT, N>::n; i++) s += (i >= o && i < o + N) ? samp(r, i) * (unsigned)(i - o + 1) : 0u;
  return s;
}
#define PIN(x) asm volatile("" : "+v"(x) : : "memory")

template <int LAYOUT, bool CHROMA, bool STORE>
__global__ void __launch_bounds__(256, 3) k_model(Planes P, const Job* jobs, int njobs, unsigned* sums, int lds_pad)
{
  extern __shared__ unsigned s_pad[];
  if (lds_pad < 0) s_pad[threadIdx.x] = 0;                                   // (keeps the allocation)
  const int nb = gridDim.x, b = blockIdx.x, xcd = b & 7, slot = b >> 3, blk = xcd * (nb >> 3) + slot;
  const int ji = blk * 256 + threadIdx.x;
  if (ji >= njobs) return;
  const Job j = jobs[ji];
  unsigned acc = 0, accc = 0;
#pragma unroll 1
  for (int l = 0; l < 2; l++) {
    const int xa = j.x0 + j.mv[l][0] - 3, ya = j.y0 + j.mv[l][1] - 3;
    const G unsigned char* pl = (const G unsigned char*)P.ref[l][0];
    /* rows in groups of three, two groups requested ahead of the one being summed (the product: row pairs, two pairs ahead) */
    constexpr int NR = RowRegs<LAYOUT, 11>::n;
    unsigned R[3][3][NR];
#pragma unroll
    for (int g = 0; g < 2; g++)
#pragma unroll
      for (int r = 0; r < 3; r++) load_row<LAYOUT, 11>(pl, W, xa, ya + 3 * g + r, R[g][r]);
#pragma unroll
    for (int g = 0; g < 5; g++) {
      if (g + 2 < 5) {
#pragma unroll
        for (int r = 0; r < 3; r++) load_row<LAYOUT, 11>(pl, W, xa, ya + 3 * (g + 2) + r, R[(g + 2) % 3][r]);
      }
#pragma unroll
      for (int r = 0; r < 3; r++) acc = acc * 31u + sum_row<LAYOUT, 11>(R[g % 3][r], xa);
      PIN(acc);
    }
    if (CHROMA) {
      const int xc = (j.x0 >> 1) + (j.mv[l][0] >> 1) - 1, yc = (j.y0 >> 1) + (j.mv[l][1] >> 1) - 1;
      constexpr int NC = RowRegs<LAYOUT, 5>::n;
      if (LAYOUT == 0) {
        unsigned C[2][7][NC];                                                 // both planes' rows in flight together (the product does the same)
#pragma unroll
        for (int c = 0; c < 2; c++)
#pragma unroll
          for (int r = 0; r < 7; r++) load_row<LAYOUT, 5>((const G unsigned char*)P.ref[l][1 + c], W / 2, xc, yc + r, C[c][r]);
#pragma unroll
        for (int c = 0; c < 2; c++)
#pragma unroll
          for (int r = 0; r < 7; r++) accc += sum_row<LAYOUT, 5>(C[c][r], xc) * (unsigned)(1 + r + 8 * c + 16 * l);
      } else {
        /* six registers per row: rows {0,1,2} {3,4} {5,6} of both planes, one group requested ahead */
        unsigned C[2][2][3][NC];
        constexpr int g0[4] = {0, 3, 5, 7};
#pragma unroll
        for (int c = 0; c < 2; c++)
#pragma unroll
          for (int r = g0[0]; r < g0[1]; r++) load_row<LAYOUT, 5>((const G unsigned char*)P.ref[l][1 + c], W / 2, xc, yc + r, C[0][c][r - g0[0]]);
#pragma unroll
        for (int g = 0; g < 3; g++) {
          if (g + 1 < 3) {
#pragma unroll
            for (int c = 0; c < 2; c++)
#pragma unroll
              for (int r = g0[g + 1]; r < g0[g + 2]; r++) load_row<LAYOUT, 5>((const G unsigned char*)P.ref[l][1 + c], W / 2, xc, yc + r, C[(g + 1) & 1][c][r - g0[g + 1]]);
          }
#pragma unroll
          for (int c = 0; c < 2; c++)
#pragma unroll
            for (int r = g0[g]; r < g0[g + 1]; r++) accc += sum_row<LAYOUT, 5>(C[g & 1][c][r - g0[g]], xc) * (unsigned)(1 + r + 8 * c + 16 * l);
          PIN(accc);
        }
      }
      PIN(accc);
    }
  }
  if (STORE) {
    unsigned char* d = P.dst[0] + (size_t)j.y0 * (W * 2) + (size_t)j.x0 * 2;
#pragma unroll
    for (int r = 0; r < 8; r++) __builtin_nontemporal_store(u2{acc + r, acc ^ r}, (u2*)(d + (size_t)r * (W * 2)));
    if (CHROMA) {
#pragma unroll
      for (int c = 0; c < 2; c++) {
        unsigned char* dc = P.dst[1 + c] + (size_t)(j.y0 >> 1) * W + (size_t)(j.x0 >> 1) * 2;
#pragma unroll
        for (int r = 0; r < 4; r++) __builtin_nontemporal_store(accc + r + c, (unsigned*)(dc + (size_t)r * W));
      }
    }
  }
  sums[ji] = acc ^ (accc * 2654435761u);
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

static void make_plane(std::vector<unsigned short>& lin, std::vector<unsigned short>& til, int pw, int ph, unsigned seed)
{
  const int phs = (ph + 7) & ~7;
  lin.assign((size_t)pw * phs + 64, 0); til.assign((size_t)pw * phs + 256, 0);
  for (int y = 0; y < ph; y++)
    for (int x = 0; x < pw; x++) {
      const unsigned short v = (unsigned short)((((unsigned)x * 2654435761u) ^ ((unsigned)y * 40503u) ^ seed) >> 7 & 0x3FF);
      lin[(size_t)y * pw + x] = v;
      til[(((size_t)(y >> 3) * (pw >> 3) + (x >> 3)) * 8 + (y & 7)) * 8 + (x & 7)] = v;
    }
}

template <int LAYOUT, bool CHROMA, bool STORE>
static float run(const Planes& P, const Job* dj, int njobs, unsigned* dsums, int lds, int reps)
{
  const int nblk = (((njobs + 255) / 256 + 7) / 8) * 8;
  hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  std::vector<float> t;
  CHK(hipFuncSetAttribute((const void*)k_model<LAYOUT, CHROMA, STORE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
  for (int r = 0; r < reps + 1; r++) {
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_model<LAYOUT, CHROMA, STORE>), dim3(nblk), dim3(256), lds, 0, P, dj, njobs, dsums, 0);
    CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    if (r) t.push_back(ms);
  }
  std::sort(t.begin(), t.end());
  CHK(hipEventDestroy(e0)); CHK(hipEventDestroy(e1));
  return t[t.size() / 2];
}

int main(int argc, char** argv)
{
  const int wg_per_cu = argc > 1 ? atoi(argv[1]) : 3;
  const int lds = wg_per_cu >= 8 ? 0 : (160 * 1024 / wg_per_cu - 1024) & ~255;          // dynamic LDS that leaves room for wg_per_cu workgroups
  // planes: two references x three planes in both layouts, one destination
  Planes PL, PT;
  std::vector<unsigned short> lin, til;
  for (int l = 0; l < 2; l++)
    for (int c = 0; c < 3; c++) {
      const int pw = c ? W / 2 : W, ph = c ? H / 2 : H;
      make_plane(lin, til, pw, ph, 77u * l + 13u * c);
      unsigned char *a, *b;
      CHK(hipMalloc(&a, lin.size() * 2)); CHK(hipMemcpy(a, lin.data(), lin.size() * 2, hipMemcpyHostToDevice));
      CHK(hipMalloc(&b, til.size() * 2)); CHK(hipMemcpy(b, til.data(), til.size() * 2, hipMemcpyHostToDevice));
      PL.ref[l][c] = a; PT.ref[l][c] = b;
    }
  for (int c = 0; c < 3; c++) { unsigned char* d; CHK(hipMalloc(&d, (size_t)W * (H + 8) * 2)); PL.dst[c] = d; PT.dst[c] = d; }

  const int mixes[][4] = {{8, 16, 32, 64}, {64, 64, 64, 64}, {32, 32, 32, 32}, {16, 16, 16, 16}, {8, 8, 8, 8}};
  const char* mixname[] = {"mix 8/16/32/64 per CTB (the bench workload's)", "64x64 only", "32x32 only", "16x16 only", "8x8 only"};
  printf("wg_per_cu %d (dynamic LDS %d B)\n", wg_per_cu, lds);
  for (int m = 0; m < 5; m++) {
    std::vector<Job> jobs;
    rnd_state = 12345;
    for (int cy = 0; cy < H; cy += 64) for (int cx = 0; cx < W; cx += 64) gen_pbs(jobs, cx, cy, 64, mixes[m][rnd() & 3]);
    const int njobs = (int)jobs.size();
    Job* dj; CHK(hipMalloc(&dj, (size_t)njobs * sizeof(Job))); CHK(hipMemcpy(dj, jobs.data(), (size_t)njobs * sizeof(Job), hipMemcpyHostToDevice));
    unsigned *s0, *s1; CHK(hipMalloc(&s0, (size_t)njobs * 4)); CHK(hipMalloc(&s1, (size_t)njobs * 4));
    // check: the tiled addressing fetches the same samples
    run<0, true, false>(PL, dj, njobs, s0, lds, 1); run<1, true, false>(PT, dj, njobs, s1, lds, 1);
    std::vector<unsigned> h0(njobs), h1(njobs);
    CHK(hipMemcpy(h0.data(), s0, (size_t)njobs * 4, hipMemcpyDeviceToHost)); CHK(hipMemcpy(h1.data(), s1, (size_t)njobs * 4, hipMemcpyDeviceToHost));
    int bad = 0; for (int i = 0; i < njobs; i++) bad += h0[i] != h1[i];
    // host check of a few jobs against the linear plane of reference 0 / luma is implied by the equality of two independent addressings
    const double alg = (double)njobs * 2 * (15 * 11 + 2 * 7 * 5) * 2 + (double)njobs * 48 * 2;
    printf("%-46s %8d jobs, layouts agree: %s\n", mixname[m], njobs, bad ? "NO" : "yes");
    const int R = 7;
    const float a0 = run<0, false, false>(PL, dj, njobs, s0, lds, R), a1 = run<1, false, false>(PT, dj, njobs, s1, lds, R);
    const float b0 = run<0, true, false>(PL, dj, njobs, s0, lds, R), b1 = run<1, true, false>(PT, dj, njobs, s1, lds, R);
    const float c0 = run<0, true, true>(PL, dj, njobs, s0, lds, R), c1 = run<1, true, true>(PT, dj, njobs, s1, lds, R);
    printf("    luma loads only          linear %.4f ms   line tiles %.4f ms   (x%.2f)\n", a0, a1, a0 / a1);
    printf("    luma + chroma loads      linear %.4f ms   line tiles %.4f ms   (x%.2f)\n", b0, b1, b0 / b1);
    printf("    loads + stores (k_inter) linear %.4f ms   line tiles %.4f ms   (x%.2f)   %.0f MB algorithmic -> %.2f / %.2f TB/s\n", c0, c1, c0 / c1, alg / 1e6, alg / c0 / 1e9, alg / c1 / 1e9);
    CHK(hipFree(dj)); CHK(hipFree(s0)); CHK(hipFree(s1));
  }
  return 0;
}
