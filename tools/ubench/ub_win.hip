// What would a 2-D tiled frame layout buy the motion-compensation fetch?  (MI355X; evidence for DESIGN.md "frame layout".)
// Every wave fetches the reference window of a 64x32 luma region displaced by a random motion vector: 39 rows x 71 samples of
// 16 bit, lanes = 16-byte chunks (so all loads are 16-byte aligned in both layouts and the instruction count is the same).
//   layout 0: linear planes, pitch 15360 B (8K 10-bit)       layout 1: tiles of 32 x 8 samples (512 B contiguous), tile-row major
// The frame is 8K (66 MB per plane, two planes used alternately) so the traffic comes from HBM / the fabric, as in k_inter.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef unsigned u4 __attribute__((ext_vector_type(4)));
#define G __attribute__((address_space(1)))
#define W 7680
#define H 4320

__device__ __forceinline__ size_t addr(int layout, int x8, int y)   // byte address of the 8-sample chunk x8 of row y
{
  if (layout == 0) return (size_t)y * (W * 2) + (size_t)x8 * 16;
  const int tx = x8 >> 2, ty = y >> 3;                                // tile 32 x 8 samples = 4 chunks x 8 rows
  return ((size_t)ty * (W / 32) + tx) * 512 + (size_t)(y & 7) * 64 + (size_t)(x8 & 3) * 16;
}

template <int LAYOUT>
__global__ void __launch_bounds__(256) k_win(unsigned* out, const unsigned char* f0, const unsigned char* f1, int nregions, unsigned seed)
{
  const int lane = threadIdx.x & 63, wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wave >= nregions) return;
  // region (64 x 32) in raster order; XCD-contiguous like k_inter: block b -> XCD b % 8 gets a contiguous eighth
  const int nb = gridDim.x, b = blockIdx.x, xcd = b & 7, slot = b >> 3, blk = xcd * (nb >> 3) + slot;
  const int region = blk * 4 + (threadIdx.x >> 6);
  const int rx = region % (W / 64), ry = region / (W / 64);
  unsigned s = seed ^ (unsigned)(region * 2654435761u); s ^= s << 13; s ^= s >> 17; s ^= s << 5;
  const int mvx = (int)(s & 31) - 16, mvy = (int)((s >> 8) & 31) - 16;
  const unsigned char* f = (s >> 20) & 1 ? f1 : f0;
  const int x0 = min(max(rx * 64 + mvx - 3, 0), W - 80), y0 = min(max(ry * 32 + mvy - 3, 0), H - 40);
  const int c0 = x0 >> 3;                       // first 8-sample chunk; the window spans 10 chunks (71 samples + alignment)
  unsigned acc = 0;
  // 39 rows x 10 chunks = 390 chunk loads by 64 lanes: 7 rounds
#pragma unroll
  for (int i = 0; i < 7; i++) {
    const int idx = i * 64 + lane;
    if (idx < 390) {
      const int r = idx / 10, c = idx - r * 10;
      const u4 v = *(const G u4*)((const G unsigned char*)f + addr(LAYOUT, c0 + c, y0 + r));
      acc += v.x ^ v.y ^ v.z ^ v.w;
    }
  }
  out[wave * 64 + lane] = acc;
}

// k_inter's own request shape on the same windows: lane = 4-sample strip x 8-row block of the region (16 strips x 4 row blocks),
// 15 window rows per lane, each row as one 16-byte + one 8-byte load at the lane's own (dword-aligned) position: neighbouring lanes
// overlap by 2/3 horizontally and 7/15 vertically and leave the de-duplication to the L1.
typedef unsigned u2 __attribute__((ext_vector_type(2)));
__global__ void __launch_bounds__(256) k_win_jobs(unsigned* out, const unsigned char* f0, const unsigned char* f1, int nregions, unsigned seed)
{
  const int lane = threadIdx.x & 63, wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wave >= nregions) return;
  const int nb = gridDim.x, b = blockIdx.x, xcd = b & 7, slot = b >> 3, blk = xcd * (nb >> 3) + slot;
  const int region = blk * 4 + (threadIdx.x >> 6);
  const int rx = region % (W / 64), ry = region / (W / 64);
  unsigned s = seed ^ (unsigned)(region * 2654435761u); s ^= s << 13; s ^= s >> 17; s ^= s << 5;
  const int mvx = (int)(s & 31) - 16, mvy = (int)((s >> 8) & 31) - 16;
  const unsigned char* f = (s >> 20) & 1 ? f1 : f0;
  const int x0 = min(max(rx * 64 + mvx - 3, 0), W - 80), y0 = min(max(ry * 32 + mvy - 3, 0), H - 40);
  const int strip = lane & 15, rblk = lane >> 4;
  const G unsigned char* base = (const G unsigned char*)f + (size_t)(y0 + 8 * rblk) * (W * 2) + (size_t)((x0 + 4 * strip) & ~1) * 2;
  unsigned acc = 0;
#pragma unroll
  for (int r = 0; r < 15; r++) {
    const u4 v = *(const G u4*)(base + (size_t)r * (W * 2));
    const u2 w = *(const G u2*)(base + (size_t)r * (W * 2) + 16);
    acc += v.x ^ v.y ^ v.z ^ v.w ^ w.x ^ w.y;
  }
  out[wave * 64 + lane] = acc;
}

// The whole memory traffic of k_inter for an 8K 4:2:0 16-bit picture made of 64x32 regions, nothing else: per lane (= 4x8 luma job)
// 15 luma window rows (16 + 8 B) and 2 x 7 chroma window rows (12 B) per list, half of the regions bi-predicted (second list from the
// other reference), then the stores of the job: 8 luma rows x 8 B, 2 x 4 chroma rows x 4 B.  No records, no arithmetic.
typedef unsigned u3 __attribute__((ext_vector_type(3), aligned(4)));
struct Planes { const unsigned char* r[2][3]; unsigned char* d[3]; };
template <int PARTS>   // bit 0 luma loads, 1 chroma loads, 2 luma stores, 3 chroma stores
__global__ void __launch_bounds__(256) k_win_full(unsigned* out, Planes P, int nregions, unsigned seed)
{
  const int lane = threadIdx.x & 63, wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wave >= nregions) return;
  const int nb = gridDim.x, b = blockIdx.x, xcd = b & 7, slot = b >> 3, blk = xcd * (nb >> 3) + slot;
  const int region = blk * 4 + (threadIdx.x >> 6);
  const int rx = region % (W / 64), ry = region / (W / 64);
  unsigned s = seed ^ (unsigned)(region * 2654435761u); s ^= s << 13; s ^= s >> 17; s ^= s << 5;
  const int npass = ((s >> 21) & 1) ? 2 : 1;
  const int strip = lane & 15, rblk = lane >> 4;
  unsigned acc = 0;
  for (int pass = 0; pass < npass; pass++) {
    const int mvx = (int)((s >> (pass * 5)) & 31) - 16, mvy = (int)((s >> (10 + pass * 5)) & 31) - 16;
    const int ref = ((s >> 20) + pass) & 1;
    const int x0 = min(max(rx * 64 + mvx - 3, 0), W - 80), y0 = min(max(ry * 32 + mvy - 3, 0), H - 40);
    if (PARTS & 1) {
      const G unsigned char* base = (const G unsigned char*)P.r[ref][0] + (size_t)(y0 + 8 * rblk) * (W * 2) + (size_t)((x0 + 4 * strip) & ~1) * 2;
#pragma unroll
      for (int r = 0; r < 15; r++) {
        const u4 v = *(const G u4*)(base + (size_t)r * (W * 2));
        const u2 w = *(const G u2*)(base + (size_t)r * (W * 2) + 16);
        acc += v.x ^ v.y ^ v.z ^ v.w ^ w.x ^ w.y;
      }
    }
    if (PARTS & 2) {
#pragma unroll
      for (int c = 1; c < 3; c++) {
        const G unsigned char* base = (const G unsigned char*)P.r[ref][c] + (size_t)(y0 / 2 + 4 * rblk) * W + (size_t)((x0 / 2 + 2 * strip) & ~1) * 2;
#pragma unroll
        for (int r = 0; r < 7; r++) { const u3 v = *(const G u3*)(base + (size_t)r * W); acc += v.x ^ v.y ^ v.z; }
      }
    }
  }
  if (PARTS & 4) {
    G unsigned char* d = (G unsigned char*)P.d[0] + (size_t)(ry * 32 + 8 * rblk) * (W * 2) + (size_t)(rx * 64 + 4 * strip) * 2;
#pragma unroll
    for (int r = 0; r < 8; r++) *(G u2*)(d + (size_t)r * (W * 2)) = (u2){acc + r, acc ^ (unsigned)r};
  }
  if (PARTS & 8) {
#pragma unroll
    for (int c = 1; c < 3; c++) {
      G unsigned char* d = (G unsigned char*)P.d[c] + (size_t)(ry * 16 + 4 * rblk) * W + (size_t)(rx * 32 + 2 * strip) * 2;
#pragma unroll
      for (int r = 0; r < 4; r++) *(G unsigned*)(d + (size_t)r * W) = acc + r + c;
    }
  }
  if (!(PARTS & 12)) out[wave * 64 + lane] = acc;
}

int main()
{
  const size_t plane = (size_t)W * H * 2;
  unsigned char *f0, *f1; CHK(hipMalloc(&f0, plane + 65536)); CHK(hipMalloc(&f1, plane + 65536));
  CHK(hipMemset(f0, 1, plane)); CHK(hipMemset(f1, 2, plane));
  const int nregions = (W / 64) * (H / 32 - 1);              // 16080 waves
  const int blocks = ((nregions + 3) / 4 + 7) / 8 * 8;
  unsigned* out; CHK(hipMalloc(&out, (size_t)blocks * 256 * 4));
  hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  for (int layout = 0; layout < 2; layout++)
    for (int rep = 0; rep < 3; rep++) {
      float ms;
      CHK(hipEventRecord(e0));
      for (int k = 0; k < 10; k++) {
        if (layout == 0) hipLaunchKernelGGL(k_win<0>, dim3(blocks), dim3(256), 0, 0, out, f0, f1, nregions, 1234u + k);
        else hipLaunchKernelGGL(k_win<1>, dim3(blocks), dim3(256), 0, 0, out, f0, f1, nregions, 1234u + k);
      }
      CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1)); CHK(hipEventElapsedTime(&ms, e0, e1));
      const double bytes = (double)nregions * 390 * 16 * 10;
      if (rep == 2) printf("layout %s: %.3f ms per launch, %.2f TB/s requested (%.0f MB per launch)\n", layout ? "tiled 32x8 " : "linear     ", ms / 10, bytes / ms / 1e9, bytes / 10 / 1e6);
    }
  // the same at limited occupancy: dynamic LDS caps the workgroups per CU (160 KB / lds) — k_inter runs 2-3 workgroups per CU
  const int ldss[5] = {0, 20 * 1024, 40 * 1024, 52 * 1024, 80 * 1024};
  for (int li = 0; li < 5; li++)
  for (int rep = 0; rep < 3; rep++) {
    float ms;
    CHK(hipEventRecord(e0));
    for (int k = 0; k < 10; k++) hipLaunchKernelGGL(k_win_jobs, dim3(blocks), dim3(256), ldss[li], 0, out, f0, f1, nregions, 1234u + k);
    CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1)); CHK(hipEventElapsedTime(&ms, e0, e1));
    const double uniq = (double)nregions * 390 * 16 * 10, req = (double)nregions * 64 * 15 * 24 * 10;
    if (rep == 2) printf("job-shaped requests (k_inter), %d workgroups/CU: %.3f ms per launch = %.2f TB/s of distinct window bytes (%.2f TB/s requested by the lanes)\n", ldss[li] ? 160 * 1024 / ldss[li] : 8, ms / 10, uniq / ms / 1e9, req / ms / 1e9);
  }
  {
    Planes P;
    for (int r = 0; r < 2; r++) { P.r[r][0] = r ? f1 : f0; for (int c = 1; c < 3; c++) { unsigned char* q; CHK(hipMalloc(&q, plane / 4 + 65536)); CHK(hipMemset(q, 3, plane / 4)); P.r[r][c] = q; } }
    CHK(hipMalloc(&P.d[0], plane + 65536)); for (int c = 1; c < 3; c++) CHK(hipMalloc(&P.d[c], plane / 4 + 65536));
    const char* names[6] = {"luma loads", "chroma loads", "all loads", "all stores", "luma loads + stores", "everything"};
    const int parts[6] = {1, 2, 3, 12, 5, 15};
    for (int v = 0; v < 6; v++)
      for (int rep = 0; rep < 3; rep++) {
        float ms;
        CHK(hipEventRecord(e0));
        for (int k = 0; k < 10; k++) {
          switch (parts[v]) {
            case 1: hipLaunchKernelGGL(k_win_full<1>, dim3(blocks), dim3(256), 0, 0, out, P, nregions, 77u + k); break;
            case 2: hipLaunchKernelGGL(k_win_full<2>, dim3(blocks), dim3(256), 0, 0, out, P, nregions, 77u + k); break;
            case 3: hipLaunchKernelGGL(k_win_full<3>, dim3(blocks), dim3(256), 0, 0, out, P, nregions, 77u + k); break;
            case 12: hipLaunchKernelGGL(k_win_full<12>, dim3(blocks), dim3(256), 0, 0, out, P, nregions, 77u + k); break;
            case 5: hipLaunchKernelGGL(k_win_full<5>, dim3(blocks), dim3(256), 0, 0, out, P, nregions, 77u + k); break;
            default: hipLaunchKernelGGL(k_win_full<15>, dim3(blocks), dim3(256), 0, 0, out, P, nregions, 77u + k); break;
          }
        }
        CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1)); CHK(hipEventElapsedTime(&ms, e0, e1));
        if (rep == 2) printf("k_inter traffic model, %-20s: %.3f ms per launch\n", names[v], ms / 10);
      }
  }
  return 0;
}
