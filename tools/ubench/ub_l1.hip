// L1 / texture-addresser throughput of the load shapes k_inter could use (MI355X): every wave of a full-occupancy
// launch re-reads a small L1/L2-resident region, so the time is the TA/TCP processing rate, not HBM.
//   mode 0: dwordx4 per lane, lane stride 8 B  (k_inter luma row part 1)   mode 1: dwordx2, stride 8 B (part 2)
//   mode 2: dwordx4, lane stride 16 B (contiguous)                          mode 3: dwordx2, stride 8 B contiguous-own (DPP variant)
//   mode 4: dwordx3, stride 8 B                                             mode 5: dword, stride 4 B
// shift = byte misalignment of the wave's base (0, 2, 4, 8)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef unsigned u4 __attribute__((ext_vector_type(4), aligned(1)));
typedef unsigned u3 __attribute__((ext_vector_type(3), aligned(1)));
typedef unsigned u2 __attribute__((ext_vector_type(2), aligned(1)));
typedef unsigned u1 __attribute__((aligned(1)));
#define G __attribute__((address_space(1)))

template <int MODE>
__global__ void __launch_bounds__(256) k_l1(unsigned* out, const unsigned char* src, int pitch, int rows, int iters, int shift)
{
  const int lane = threadIdx.x & 63, wave = (blockIdx.x * 4 + (threadIdx.x >> 6));
  const int stride = MODE == 2 ? 16 : (MODE == 5 ? 4 : 8);
  const G unsigned char* base = (const G unsigned char*)src + (size_t)(wave % 61) * 4096 + lane * stride + shift;
  unsigned acc = 0;
  for (int it = 0; it < iters; it++) {
#pragma unroll 8
    for (int r = 0; r < rows; r++) {
      const G unsigned char* p = base + (size_t)r * pitch;
      if (MODE == 0 || MODE == 2) { const u4 v = *(const G u4*)p; acc += v.x ^ v.y ^ v.z ^ v.w; }
      else if (MODE == 1 || MODE == 3) { const u2 v = *(const G u2*)p; acc += v.x ^ v.y; }
      else if (MODE == 4) { const u3 v = *(const G u3*)p; acc += v.x ^ v.y ^ v.z; }
      else { acc += *(const G u1*)p; }
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main()
{
  const int pitch = 16384, rows = 16, iters = 200, blocks = 256 * 8;
  unsigned char* src; CHK(hipMalloc(&src, (size_t)pitch * 64 + 61 * 4096 + 65536)); CHK(hipMemset(src, 1, (size_t)pitch * 64 + 61 * 4096 + 65536));
  unsigned* out; CHK(hipMalloc(&out, blocks * 256 * 4));
  hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  const char* names[6] = {"dwordx4 stride8", "dwordx2 stride8", "dwordx4 stride16", "dwordx2 stride8(b)", "dwordx3 stride8", "dword stride4"};
  const int bytes[6] = {16, 8, 16, 8, 12, 4};
  for (int mode = 0; mode < 6; mode++)
    for (int shift = 0; shift <= 8; shift = shift ? shift * 2 : 2) {
      float ms = 0;
      for (int rep = 0; rep < 2; rep++) {
        CHK(hipEventRecord(e0));
        switch (mode) {
          case 0: hipLaunchKernelGGL(k_l1<0>, dim3(blocks), dim3(256), 0, 0, out, src, pitch, rows, iters, shift); break;
          case 1: hipLaunchKernelGGL(k_l1<1>, dim3(blocks), dim3(256), 0, 0, out, src, pitch, rows, iters, shift); break;
          case 2: hipLaunchKernelGGL(k_l1<2>, dim3(blocks), dim3(256), 0, 0, out, src, pitch, rows, iters, shift); break;
          case 3: hipLaunchKernelGGL(k_l1<3>, dim3(blocks), dim3(256), 0, 0, out, src, pitch, rows, iters, shift); break;
          case 4: hipLaunchKernelGGL(k_l1<4>, dim3(blocks), dim3(256), 0, 0, out, src, pitch, rows, iters, shift); break;
          default: hipLaunchKernelGGL(k_l1<5>, dim3(blocks), dim3(256), 0, 0, out, src, pitch, rows, iters, shift); break;
        }
        CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1)); CHK(hipEventElapsedTime(&ms, e0, e1));
      }
      const double ninstr = (double)blocks * 4 * rows * iters;                 // wave-level load instructions
      printf("%-20s shift %d: %.3f ms  %.2f G wave-loads/s  %.1f cycles/instr/CU @2.4GHz  %.2f TB/s requested\n", names[mode], shift, ms,
             ninstr / ms / 1e6, 2.4e9 * (ms * 1e-3) / (ninstr / 256), ninstr * 64 * bytes[mode] / ms / 1e9);
    }
  return 0;
}
