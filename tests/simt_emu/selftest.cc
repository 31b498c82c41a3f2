/* tests/simt_emu/selftest.cc — TEST INFRASTRUCTURE: does the SIMT interpreter catch what it is meant to catch?
 * Three deliberately WRONG kernels (a missing barrier between waves, a block that reads its neighbour block's result, a read of
 * memory nobody wrote) must give wrong answers here, as they would — sooner or later — on the GPU; their correct twins must not.
 * Built and run by tests/test_emu_selftest.py:  g++ -x c++ -I tests/simt_emu selftest.cc simt_emu.cpp */
#include <hip/hip_runtime.h>

/* every wave writes its slot, then wave w reads the slot of wave w-1 (wave 0 its own): needs a barrier in between — and goes
 * unnoticed for as long as the waves happen to run in index order */
template <bool BARRIER> __global__ void k_wave_neighbour(int* out)
{
  __shared__ int s[4];
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) s[w] = 100 + w;
  if (BARRIER) __syncthreads();
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + w] = s[w ? w - 1 : 0];
  __syncthreads();
  if ((threadIdx.x & 63) == 0) s[w] = -1;      /* the next block of the launch must not find this block's values */
}

/* block b adds the result of block b-1: only right if the blocks run in index order — which nothing promises */
__global__ void k_block_chain(int* acc)
{
  if (threadIdx.x == 0) acc[blockIdx.x + 1] = acc[blockIdx.x] + 1;
}

/* each block only touches its own element: right in any order */
__global__ void k_block_own(int* acc)
{
  if (threadIdx.x == 0) acc[blockIdx.x + 1] = (int)blockIdx.x + 1;
}

int main()
{
  int fails = 0;
  int* d = nullptr;
  hipMalloc((void**)&d, 4096 * sizeof(int));
  /* 1: fresh memory is not zero */
  if (d[0] == 0 && d[4095] == 0) { printf("FAIL: fresh device memory reads as zero\n"); fails++; }
  /* 2: waves */
  int bad = 0, good = 0;
  for (int rep = 0; rep < 8; rep++) {
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_wave_neighbour<false>), dim3(16), dim3(256), 0, 0, d);
    for (int i = 0; i < 64; i++) bad += d[i] != 100 + ((i & 3) ? (i & 3) - 1 : 0);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_wave_neighbour<true>), dim3(16), dim3(256), 0, 0, d);
    for (int i = 0; i < 64; i++) good += d[i] != 100 + ((i & 3) ? (i & 3) - 1 : 0);
  }
  if (!bad) { printf("FAIL: the missing barrier between waves went unnoticed\n"); fails++; }
  if (good) { printf("FAIL: the kernel with the barrier gave %d wrong values\n", good); fails++; }
  /* 3: blocks */
  bad = good = 0;
  for (int rep = 0; rep < 4; rep++) {
    for (int i = 0; i <= 64; i++) d[i] = 0;
    hipLaunchKernelGGL(k_block_chain, dim3(64), dim3(64), 0, 0, d);
    bad += d[64] != 64;
    for (int i = 0; i <= 64; i++) d[i] = 0;
    hipLaunchKernelGGL(k_block_own, dim3(64), dim3(64), 0, 0, d);
    for (int i = 1; i <= 64; i++) good += d[i] != i;
  }
  if (!bad) { printf("FAIL: blocks ran in index order every time\n"); fails++; }
  if (good) { printf("FAIL: order-independent blocks gave %d wrong values\n", good); fails++; }
  hipFree(d);
  printf(fails ? "simt_emu selftest: %d FAILED\n" : "simt_emu selftest ok\n", fails);
  return fails ? 1 : 0;
}
