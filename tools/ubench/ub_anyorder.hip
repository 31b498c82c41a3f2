// Does hipExtAnyOrderLaunch (an AQL dispatch packet WITHOUT the barrier bit) work on this stack (gfx950, ROCm 7.2)?  hip_ext.h says "not supported on AMD GFX9xx
// boards" for the module form.  If it does, two launches on ONE stream can overlap without a second stream and its fork / join events:
//   [P][A][B any-order][C]:  A waits for P (barrier bit), B starts as soon as the command processor has launched A, C waits for A and B.
// Checks: (1) two 500 us spin kernels back to back take ~500 us with the flag, ~1000 us without; (2) C sees the end flags of A and B; (3) B does not start
// before P has ended (the queue is processed in order: B's packet is only looked at after A's barrier is satisfied).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <stdlib.h>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// flags[0..3]: end flags of P, A, B; what B saw of P's flag at its start; what C saw of A's and B's
__global__ void k_spin(unsigned* flags, int me, int us, int look_at, int store_at)
{
  if (look_at >= 0 && blockIdx.x == 0 && threadIdx.x == 0) flags[store_at] = __hip_atomic_load(&flags[look_at], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;   // 1: not set, 2: set
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < (unsigned long long)us * 100ull) __builtin_amdgcn_s_sleep(8);     // wall_clock64: 100 MHz
  if (blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(&flags[me], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ void k_check(unsigned* flags, int a, int b, int store_at)
{
  if (blockIdx.x == 0 && threadIdx.x == 0)
    flags[store_at] = __hip_atomic_load(&flags[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) * 2 + __hip_atomic_load(&flags[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 10;   // 13: both ended
}

int main()
{
  hipStream_t st; CHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  unsigned* flags; CHK(hipMalloc(&flags, 64));
  hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  const dim3 grid(128), block(64);     // half the CUs: both kernels fit side by side
  for (int any = 0; any < 2; any++) {
    for (int rep = 0; rep < 3; rep++) {
      CHK(hipMemsetAsync(flags, 0, 64, st));
      CHK(hipStreamSynchronize(st));
      CHK(hipEventRecord(e0, st));
      hipLaunchKernelGGL(k_spin, grid, block, 0, st, flags, 0, 200, -1, 0);                                  // P
      hipLaunchKernelGGL(k_spin, grid, block, 0, st, flags, 1, 500, -1, 0);                                  // A
      hipExtLaunchKernelGGL(k_spin, grid, block, 0, st, nullptr, nullptr, any ? hipExtAnyOrderLaunch : 0, flags, 2, 500, 0, 3);   // B: looks at P's flag
      hipLaunchKernelGGL(k_check, dim3(1), dim3(64), 0, st, flags, 1, 2, 4);                                 // C
      CHK(hipEventRecord(e1, st)); CHK(hipEventSynchronize(e1));
      CHK(hipGetLastError());
      float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
      unsigned h[8]; CHK(hipMemcpy(h, flags, 32, hipMemcpyDeviceToHost));
      printf("%s  P 200 us + A 500 us + B 500 us + check: %.3f ms   B saw P ended: %s   C saw A and B ended: %s\n", any ? "B any-order" : "B in order ", ms,
             h[3] == 2 ? "yes" : (h[3] == 1 ? "NO" : "?"), h[4] == 13 ? "yes" : "NO");
    }
  }
  return 0;
}
