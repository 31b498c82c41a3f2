// Cost of wave-level global stores as k_inter issues them (MI355X): groups of `group` lanes write adjacent 4 / 8 / 16-byte pieces
// of one row, groups sit 8 rows apart.  The destination (a few MB) stays in L2: the time is the store path's processing rate.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef unsigned u4 __attribute__((ext_vector_type(4)));
typedef unsigned u2 __attribute__((ext_vector_type(2)));
#define G __attribute__((address_space(1)))

template <int BYTES>
__global__ void __launch_bounds__(256) k_st(unsigned char* dst, int pitch, int rows, int iters, int group)
{
  const int lane = threadIdx.x & 63, wave = (blockIdx.x * 4 + (threadIdx.x >> 6));
  G unsigned char* base = (G unsigned char*)dst + (size_t)(wave % 61) * 65536 + (lane % group) * BYTES + (size_t)(lane / group) * 8 * pitch;
  for (int it = 0; it < iters; it++) {
#pragma unroll 8
    for (int r = 0; r < rows; r++) {
      G unsigned char* p = base + (size_t)r * pitch;
      if (BYTES == 16) *(G u4*)p = (u4){(unsigned)it, (unsigned)r, 3u, 4u};
      else if (BYTES == 8) *(G u2*)p = (u2){(unsigned)it, (unsigned)r};
      else *(G unsigned*)p = (unsigned)(it + r);
    }
  }
}

int main()
{
  const int pitch = 16384, rows = 8, iters = 100, blocks = 256 * 8;
  const size_t sz = (size_t)pitch * 1024 + 61 * 65536;
  unsigned char* dst; CHK(hipMalloc(&dst, sz)); CHK(hipMemset(dst, 1, sz));
  hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  struct { int bytes, group; } cfg[] = {{8, 16}, {8, 8}, {8, 4}, {8, 2}, {16, 16}, {16, 8}, {16, 4}, {4, 16}, {4, 8}, {4, 2}, {8, 64}, {16, 64}, {4, 64}};
  for (auto& c : cfg) {
    float ms = 0;
    for (int rep = 0; rep < 2; rep++) {
      CHK(hipEventRecord(e0));
      if (c.bytes == 16) hipLaunchKernelGGL(k_st<16>, dim3(blocks), dim3(256), 0, 0, dst, pitch, rows, iters, c.group);
      else if (c.bytes == 8) hipLaunchKernelGGL(k_st<8>, dim3(blocks), dim3(256), 0, 0, dst, pitch, rows, iters, c.group);
      else hipLaunchKernelGGL(k_st<4>, dim3(blocks), dim3(256), 0, 0, dst, pitch, rows, iters, c.group);
      CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1)); CHK(hipEventElapsedTime(&ms, e0, e1));
    }
    const double ninstr = (double)blocks * 4 * rows * iters;
    printf("store %2d B/lane, groups of %2d lanes: %.3f ms  %6.1f cycles/instr/CU @2.4GHz  %.2f TB/s\n", c.bytes, c.group, ms,
           2.4e9 * (ms * 1e-3) / (ninstr / 256), ninstr * 64 * c.bytes / ms / 1e9);
  }
  return 0;
}
