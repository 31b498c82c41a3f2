// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on MI355X for k_inter-like access patterns (MI355X_MICROARCH.md, HBM section:
// "other access widths and WRITE_SIZE are uncalibrated: calibrate on a known byte count in your own access pattern").
// Every kernel touches a KNOWN number of distinct bytes exactly once, in a buffer far larger than L2 + Infinity Cache:
//   rd_stream      : 16 B per lane, fully coalesced                               (the guide's x2 case)
//   rd_seg<S>      : S-byte segments (16 / 32 / 64 / 128), one per group of S/16 lanes, segments 512 B apart  (window-row fragments)
//   wr_stream      : 16 B per lane, fully coalesced
//   wr_seg<S>      : S-byte segments 512 B apart, each written once                 (PB row pieces: 16 B = an 8-wide 10-bit PB row)
//   wr_halves      : every 128-byte line written as two 64-byte halves by DIFFERENT workgroups far apart in time (the chroma case:
//                    a 4:2:0 chroma row of a 64x64 CTB is 64 B at 10 bit, the other half of the line belongs to the next CTB)
// usage: rocprofv3 --pmc FETCH_SIZE -- ./ub_pmc_cal ; rocprofv3 --pmc WRITE_SIZE -- ./ub_pmc_cal   (prints the expected bytes per kernel)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef unsigned u4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) rd_stream(const u4* src, size_t n, unsigned* sink)
{
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const u4 v = src[i];
  if ((v.x ^ v.y ^ v.z ^ v.w) == 0x12345678u) sink[0] = 1;
}
template <int S>
__global__ void __launch_bounds__(256) rd_seg(const unsigned char* src, size_t nseg, unsigned* sink)
{
  constexpr int LPS = S / 16;                         // lanes per segment
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t seg = t / LPS;
  if (seg >= nseg) return;
  const u4 v = *(const u4*)(src + seg * 512 + (t % LPS) * 16);
  if ((v.x ^ v.y ^ v.z ^ v.w) == 0x12345678u) sink[0] = 1;
}
__global__ void __launch_bounds__(256) wr_stream(u4* dst, size_t n)
{
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) dst[i] = (u4){(unsigned)i, 1u, 2u, 3u};
}
template <int S>
__global__ void __launch_bounds__(256) wr_seg(unsigned char* dst, size_t nseg)
{
  constexpr int LPS = S / 16;
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t seg = t / LPS;
  if (seg >= nseg) return;
  *(u4*)(dst + seg * 512 + (t % LPS) * 16) = (u4){(unsigned)t, 1u, 2u, 3u};
}
// first half of the grid writes the low 64 B of every line, second half the high 64 B
__global__ void __launch_bounds__(256) wr_halves(unsigned char* dst, size_t nlines)
{
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;          // 4 lanes per half line
  const size_t h = t / 4;
  if (h >= 2 * nlines) return;
  const size_t line = h % nlines, half = h / nlines;
  *(u4*)(dst + line * 128 + half * 64 + (t % 4) * 16) = (u4){(unsigned)t, 1u, 2u, 3u};
}

int main()
{
  const size_t SZ = (size_t)2 << 30;                  // 2 GiB: far beyond 32 MB of L2 and 256 MB of Infinity Cache
  unsigned char* buf; CHK(hipMalloc(&buf, SZ)); CHK(hipMemset(buf, 1, SZ));
  unsigned* sink; CHK(hipMalloc(&sink, 4));
  const size_t nseg = SZ / 512;
  CHK(hipDeviceSynchronize());
  printf("expected bytes: rd_stream %zu  rd_seg<S> S*%zu  wr_stream %zu  wr_seg<S> S*%zu  wr_halves %zu\n", SZ / 2, nseg, SZ / 2, nseg, SZ / 2);
  hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
#define RUN(name, useful, ...) do { for (int rep = 0; rep < 2; rep++) { CHK(hipEventRecord(e0)); hipLaunchKernelGGL(__VA_ARGS__); CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1)); float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); \
    if (rep) printf("%-12s %8.3f ms  %7.2f TB/s useful\n", name, ms, (double)(useful) / ms / 1e9); } } while (0)
  RUN("rd_stream", SZ / 2, rd_stream, dim3((unsigned)((SZ / 2 / 16 + 255) / 256)), dim3(256), 0, 0, (const u4*)buf, SZ / 2 / 16, sink);
  RUN("rd_seg16", nseg * 16, rd_seg<16>, dim3((unsigned)((nseg * 1 + 255) / 256)), dim3(256), 0, 0, buf, nseg, sink);
  RUN("rd_seg32", nseg * 32, rd_seg<32>, dim3((unsigned)((nseg * 2 + 255) / 256)), dim3(256), 0, 0, buf, nseg, sink);
  RUN("rd_seg64", nseg * 64, rd_seg<64>, dim3((unsigned)((nseg * 4 + 255) / 256)), dim3(256), 0, 0, buf, nseg, sink);
  RUN("rd_seg128", nseg * 128, rd_seg<128>, dim3((unsigned)((nseg * 8 + 255) / 256)), dim3(256), 0, 0, buf, nseg, sink);
  RUN("wr_stream", SZ / 2, wr_stream, dim3((unsigned)((SZ / 2 / 16 + 255) / 256)), dim3(256), 0, 0, (u4*)buf, SZ / 2 / 16);
  RUN("wr_seg16", nseg * 16, wr_seg<16>, dim3((unsigned)((nseg * 1 + 255) / 256)), dim3(256), 0, 0, buf, nseg);
  RUN("wr_seg32", nseg * 32, wr_seg<32>, dim3((unsigned)((nseg * 2 + 255) / 256)), dim3(256), 0, 0, buf, nseg);
  RUN("wr_seg64", nseg * 64, wr_seg<64>, dim3((unsigned)((nseg * 4 + 255) / 256)), dim3(256), 0, 0, buf, nseg);
  RUN("wr_seg128", nseg * 128, wr_seg<128>, dim3((unsigned)((nseg * 8 + 255) / 256)), dim3(256), 0, 0, buf, nseg);
  RUN("wr_halves", SZ / 2, wr_halves, dim3((unsigned)((SZ / 2 / 128 * 8 + 255) / 256)), dim3(256), 0, 0, buf, SZ / 2 / 128);
  return 0;
}
