// Micro-benchmarks behind k_inter's design choices (run on the MI355X box):
//  (1) throughput of v_dot2c_i32_i16 vs v_mad_i32_i24 chains,
//  (2) correctness + rate of 2-byte-aligned global_load_dwordx4 / dwordx2.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
typedef short short2_ __attribute__((ext_vector_type(2)));
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void k_dot2(int* out, int iters, int a0, int b0)
{
  int acc[8];
  short2_ a = __builtin_bit_cast(short2_, a0 + (int)threadIdx.x), b = __builtin_bit_cast(short2_, b0);
#pragma unroll
  for (int k = 0; k < 8; k++) acc[k] = k;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int k = 0; k < 8; k++) acc[k] = __builtin_amdgcn_sdot2(a, b, acc[k], false);
    a = __builtin_bit_cast(short2_, __builtin_bit_cast(int, a) ^ acc[0]);
  }
  int s = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) s += acc[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_mad(int* out, int iters, int a0, int b0)
{
  int acc[8];
  int a = a0 + (int)threadIdx.x, b = b0;
#pragma unroll
  for (int k = 0; k < 8; k++) acc[k] = k;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int k = 0; k < 8; k++) acc[k] = __mul24(a, b) + acc[k];
    a ^= acc[0] & 0xFFFF;
  }
  int s = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) s += acc[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

struct __attribute__((packed, aligned(2))) V16 { unsigned v[4]; };
struct __attribute__((packed, aligned(2))) V8 { unsigned v[2]; };
// every lane loads 24 bytes starting at a 2-byte aligned address (like a 12-sample u16 window row)
__global__ void k_unaligned(unsigned* out, const unsigned short* src, int pitch, int rows, int shift)
{
  const int lane = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned acc = 0;
  for (int r = 0; r < rows; r++) {
    const unsigned short* p = src + (size_t)r * pitch + lane * 4 + shift;
    V16 a = *(const V16*)p;
    V8 b = *(const V8*)(p + 8);
    acc += a.v[0] + 3 * a.v[1] + 5 * a.v[2] + 7 * a.v[3] + 11 * b.v[0] + 13 * b.v[1];
  }
  out[lane] = acc;
}

int main()
{
  int *d; CHK(hipMalloc(&d, 256 * 2048 * 4 * 4));
  hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  const int iters = 20000, blocks = 256 * 8, threads = 256;
  for (int which = 0; which < 2; which++) {
    for (int rep = 0; rep < 2; rep++) {
      CHK(hipEventRecord(e0));
      if (which == 0) hipLaunchKernelGGL(k_dot2, dim3(blocks), dim3(threads), 0, 0, d, iters, 0x00030004, 0x00050006);
      else hipLaunchKernelGGL(k_mad, dim3(blocks), dim3(threads), 0, 0, d, iters, 0x0304, 0x0506);
      CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
      float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
      double ops = (double)blocks * threads * iters * 8;
      if (rep) printf("%s: %.3f ms, %.2f T lane-ops/s\n", which ? "v_mad(mul_i24+add)" : "v_dot2c_i32_i16", ms, ops / ms / 1e9);
    }
  }
  // unaligned loads
  const int pitch = 8192, rows = 2048, lanes = 1920;
  std::vector<unsigned short> h((size_t)pitch * rows + 64);
  for (size_t i = 0; i < h.size(); i++) h[i] = (unsigned short)(i * 2654435761u >> 13);
  unsigned short* ds; CHK(hipMalloc(&ds, h.size() * 2)); CHK(hipMemcpy(ds, h.data(), h.size() * 2, hipMemcpyHostToDevice));
  unsigned* dout; CHK(hipMalloc(&dout, lanes * 4));
  std::vector<unsigned> got(lanes);
  for (int shift = 0; shift < 4; shift++) {
    float ms = 0;
    for (int rep = 0; rep < 2; rep++) {
      CHK(hipEventRecord(e0));
      hipLaunchKernelGGL(k_unaligned, dim3(lanes / 64), dim3(64), 0, 0, dout, ds, pitch, rows, shift);
      CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
      CHK(hipEventElapsedTime(&ms, e0, e1));
    }
    CHK(hipMemcpy(got.data(), dout, lanes * 4, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int l = 0; l < lanes; l++) {
      unsigned acc = 0;
      for (int r = 0; r < rows; r++) {
        const unsigned short* p = &h[(size_t)r * pitch + l * 4 + shift];
        unsigned v[6]; memcpy(v, p, 24);
        acc += v[0] + 3 * v[1] + 5 * v[2] + 7 * v[3] + 11 * v[4] + 13 * v[5];
      }
      if (acc != got[l]) bad++;
    }
    printf("unaligned shift %d samples: %s, %.3f ms (%.1f GB/s of requested bytes)\n", shift, bad ? "MISMATCH" : "ok", ms, (double)lanes * rows * 24 / ms / 1e6);
  }
  return 0;
}
