// L1 / texture-addresser cost of a wave-level dwordx4 load as a function of how the lanes' addresses are spread
// (k_inter: consecutive lanes = horizontally adjacent 4-sample strips of one prediction block, 8 bytes apart; a PB of width
// W gives groups of W/4 lanes per row, the next group sits 8 rows further down or in another PB altogether).
//   group = lanes per row segment (16: 64-wide PB ... 2: 8-wide PB), gap = byte distance between groups
// The data set is L2/L1 resident: the time is the TA/TCP processing rate.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef unsigned u4 __attribute__((ext_vector_type(4), aligned(1)));
typedef unsigned u2 __attribute__((ext_vector_type(2), aligned(1)));
#define G __attribute__((address_space(1)))

template <int BYTES>
__global__ void __launch_bounds__(256) k_l1(unsigned* out, const unsigned char* src, int pitch, int rows, int iters, int group, int gap, int lstride)
{
  const int lane = threadIdx.x & 63, wave = (blockIdx.x * 4 + (threadIdx.x >> 6));
  const G unsigned char* base = (const G unsigned char*)src + (size_t)(wave % 7) * 4096 + (lane % group) * lstride + (size_t)(lane / group) * gap + 4;
  unsigned acc = 0;
  for (int it = 0; it < iters; it++) {
#pragma unroll 8
    for (int r = 0; r < rows; r++) {
      const G unsigned char* p = base + (size_t)r * pitch;
      if (BYTES == 16) { const u4 v = *(const G u4*)p; acc += v.x ^ v.y ^ v.z ^ v.w; }
      else { const u2 v = *(const G u2*)p; acc += v.x ^ v.y; }
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main()
{
  const int pitch = 16384, rows = 15, iters = 100, blocks = 256 * 8;
  const size_t sz = (size_t)pitch * 1024;
  unsigned char* src; CHK(hipMalloc(&src, sz)); CHK(hipMemset(src, 1, sz));
  unsigned* out; CHK(hipMalloc(&out, blocks * 256 * 4));
  hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  struct { int bytes, group, gap, lstride; const char* what; } cfg[] = {
    {16, 64, 0, 8, "x4 64 lanes in one row, 8 B apart"}, {16, 64, 0, 16, "x4 64 lanes in one row, 16 B apart (contiguous)"},
    {16, 16, 8 * 16384, 8, "x4 groups of 16 (64-wide PB), 8 rows apart"}, {16, 8, 8 * 16384, 8, "x4 groups of 8 (32-wide PB)"},
    {16, 4, 8 * 16384 + 256, 8, "x4 groups of 4 (16-wide PB)"}, {16, 2, 8 * 16384 + 256, 8, "x4 groups of 2 (8-wide PB)"},
    {16, 16, 8 * 16384, 16, "x4 groups of 16, 16 B apart"}, {16, 4, 8 * 16384 + 256, 16, "x4 groups of 4, 16 B apart"},
    {8, 64, 0, 8, "x2 64 lanes in one row, 8 B apart"}, {8, 16, 8 * 16384, 8, "x2 groups of 16"}, {8, 2, 8 * 16384 + 256, 8, "x2 groups of 2"},
  };
  for (auto& c : cfg) {
    float ms = 0;
    for (int rep = 0; rep < 2; rep++) {
      CHK(hipEventRecord(e0));
      if (c.bytes == 16) hipLaunchKernelGGL(k_l1<16>, dim3(blocks), dim3(256), 0, 0, out, src, pitch, rows, iters, c.group, c.gap, c.lstride);
      else hipLaunchKernelGGL(k_l1<8>, dim3(blocks), dim3(256), 0, 0, out, src, pitch, rows, iters, c.group, c.gap, c.lstride);
      CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1)); CHK(hipEventElapsedTime(&ms, e0, e1));
    }
    const double ninstr = (double)blocks * 4 * rows * iters;
    printf("%-50s %.3f ms  %6.1f cycles/instr/CU @2.4GHz\n", c.what, ms, 2.4e9 * (ms * 1e-3) / (ninstr / 256));
  }
  return 0;
}
