// Issue rate and dependent-issue latency of the VALU instructions the kernels' arithmetic arguments rest on (gfx950).
//   hipcc -O3 --offload-arch=gfx950 -o ub_valu_rate tools/ubench/ub_valu_rate.hip && ./ub_valu_rate
// For every instruction: K independent dependency chains (K = 1, 2, 4, 8) in one wave, W waves per SIMD (W = 1, 2, 4, 8; 256-lane
// workgroups = one wave per SIMD, W workgroups per CU, capped by a dynamic LDS allocation nobody uses), every CU busy.
// Printed: shader cycles per instruction seen by ONE wave (s_memtime around the loop), and the SIMD's aggregate rate
// = W x instructions / cycles (wave-instructions per cycle per SIMD; 0.5 = one wave64 instruction per 2 cycles, the SIMD-32 rate
// the MI355X guide quotes for v_fma_f32, 0.25 = one per 4 cycles).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

#define ITERS 256
#define PER_ITER 32   // instructions per loop iteration (unrolled): chains are visited round-robin

// one instruction on accumulator %0 (read-write) with operands %1, %2
#define DEF_KERNEL(name, text)                                                                                         \
  template <int K> __global__ void __launch_bounds__(256) name(unsigned long long* out, unsigned a0, unsigned b0)     \
  {                                                                                                                    \
    extern __shared__ unsigned pad[];                                                                                  \
    unsigned acc[8];                                                                                                   \
    unsigned a = a0 + threadIdx.x, b = b0 ^ (threadIdx.x << 3);                                                        \
    _Pragma("unroll") for (int k = 0; k < 8; k++) acc[k] = k + threadIdx.x;                                            \
    const unsigned long long t0 = clock64();                                                                           \
    for (int i = 0; i < ITERS; i++) {                                                                                  \
      _Pragma("unroll") for (int j = 0; j < PER_ITER; j++) asm volatile(text : "+v"(acc[j % K]) : "v"(a), "v"(b));      \
    }                                                                                                                  \
    const unsigned long long t1 = clock64();                                                                           \
    unsigned s = 0;                                                                                                    \
    _Pragma("unroll") for (int k = 0; k < 8; k++) s += acc[k];                                                         \
    if (s == 0x12345678u) pad[threadIdx.x] = s;                                                                        \
    if ((threadIdx.x & 63) == 0) out[(blockIdx.x * 256 + threadIdx.x) >> 6] = t1 - t0;                                 \
  }

DEF_KERNEL(k_fma, "v_fma_f32 %0, %1, %2, %0")
DEF_KERNEL(k_dot2c, "v_dot2c_i32_i16 %0, %1, %2")
DEF_KERNEL(k_dot2, "v_dot2_i32_i16 %0, %1, %2, %0")
DEF_KERNEL(k_dot4c, "v_dot4c_i32_i8 %0, %1, %2")
DEF_KERNEL(k_dot4, "v_dot4_i32_i8 %0, %1, %2, %0")
DEF_KERNEL(k_mad24, "v_mad_i32_i24 %0, %1, %2, %0")
DEF_KERNEL(k_perm, "v_perm_b32 %0, %0, %1, %2")
DEF_KERNEL(k_alignbit, "v_alignbit_b32 %0, %0, %1, %2")
DEF_KERNEL(k_add, "v_add_u32 %0, %0, %1")
DEF_KERNEL(k_pkadd, "v_pk_add_i16 %0, %0, %1 clamp")
DEF_KERNEL(k_pkashr, "v_pk_ashrrev_i16 %0, 1, %0")
DEF_KERNEL(k_pkmax, "v_pk_max_i16 %0, %0, %1")
DEF_KERNEL(k_med3, "v_med3_i32 %0, %0, %1, %2")
DEF_KERNEL(k_bfi, "v_bfi_b32 %0, %1, %0, %2")
DEF_KERNEL(k_mov, "v_mov_b32 %0, %1")
DEF_KERNEL(k_lshladd, "v_lshl_add_u32 %0, %0, 1, %1")
DEF_KERNEL(k_mullo, "v_mul_lo_u32 %0, %0, %1")

template <class F> static void run(const char* name, int K, F launch)
{
  unsigned long long* d;
  const int cus = 256;
  CHK(hipMalloc(&d, sizeof(unsigned long long) * cus * 8 * 4));
  printf("%-18s K=%d :", name, K);
  for (int W : {1, 2, 4, 8}) {
    const unsigned lds = (160 * 1024) / W - 1024;
    const int blocks = cus * W;
    launch(blocks, lds, d);                        // warm-up
    CHK(hipDeviceSynchronize());
    launch(blocks, lds, d);
    CHK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(blocks * 4);
    CHK(hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost));
    double sum = 0;
    for (auto v : h) sum += (double)v;
    const double cyc = sum / h.size(), n = (double)ITERS * PER_ITER;
    printf("  W=%d %6.2f cyc/instr/wave (%.3f instr/cyc/SIMD)", W, cyc / n, W * n / cyc);
  }
  printf("\n");
  CHK(hipFree(d));
}

#define RUN(kern, label)                                                                                                             \
  do {                                                                                                                               \
    run(label, 1, [](int b, unsigned l, unsigned long long* d) { hipLaunchKernelGGL(kern<1>, dim3(b), dim3(256), l, 0, d, 0x00030004u, 0x00050006u); }); \
    run(label, 2, [](int b, unsigned l, unsigned long long* d) { hipLaunchKernelGGL(kern<2>, dim3(b), dim3(256), l, 0, d, 0x00030004u, 0x00050006u); }); \
    run(label, 4, [](int b, unsigned l, unsigned long long* d) { hipLaunchKernelGGL(kern<4>, dim3(b), dim3(256), l, 0, d, 0x00030004u, 0x00050006u); }); \
    run(label, 8, [](int b, unsigned l, unsigned long long* d) { hipLaunchKernelGGL(kern<8>, dim3(b), dim3(256), l, 0, d, 0x00030004u, 0x00050006u); }); \
  } while (0)

int main()
{
  hipDeviceProp_t pr;
  CHK(hipGetDeviceProperties(&pr, 0));
  printf("# %s, %d CUs, clock64() ticks; %d instructions per wave and run\n", pr.gcnArchName, pr.multiProcessorCount, ITERS * PER_ITER);
  RUN(k_fma, "v_fma_f32");
  RUN(k_dot2c, "v_dot2c_i32_i16");
  RUN(k_dot2, "v_dot2_i32_i16");
  RUN(k_dot4c, "v_dot4c_i32_i8");
  RUN(k_dot4, "v_dot4_i32_i8");
  RUN(k_mad24, "v_mad_i32_i24");
  RUN(k_mullo, "v_mul_lo_u32");
  RUN(k_perm, "v_perm_b32");
  RUN(k_alignbit, "v_alignbit_b32");
  RUN(k_add, "v_add_u32");
  RUN(k_lshladd, "v_lshl_add_u32");
  RUN(k_pkadd, "v_pk_add_i16 clamp");
  RUN(k_pkashr, "v_pk_ashrrev_i16");
  RUN(k_pkmax, "v_pk_max_i16");
  RUN(k_med3, "v_med3_i32");
  RUN(k_bfi, "v_bfi_b32");
  RUN(k_mov, "v_mov_b32");
  return 0;
}
