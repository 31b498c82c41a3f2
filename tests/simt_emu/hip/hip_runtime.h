/*
 * tests/simt_emu/hip/hip_runtime.h — KERNEL-LOGIC VERIFICATION HARNESS (test tier only).
 *
 * A single-threaded, fiber-based SIMT interpreter that lets the CPU test tier compile the product's
 * .hip sources UNCHANGED with g++ (this directory shadows <hip/hip_runtime.h>) and execute the
 * kernels block by block: every HIP thread is a fiber (simt_emu.cpp: a six-register switch on x86-64,
 * ucontext elsewhere and under AddressSanitizer), __syncthreads() and the wave-level
 * collectives (__shfl*, __ballot, wave barrier) are real rendezvous points, LDS is the kernels'
 * own static __shared__ storage, global memory is host memory.  It exists because the build
 * container has no GPU: indexing, LDS staging, barrier placement and divergence bugs are caught
 * here (a collective reached by only part of a wave is reported as a deadlock), before spending
 * scarce MI355X minutes.  It is NOT a CPU fallback: nothing in libde265_amd/ can load it, the
 * product library is always the hipcc/gfx950 build, and timing here is meaningless.
 *
 * Limits (by design): blocks run one after another in blockIdx order (so inter-block spin waits must
 * only ever wait on lower tickets — which is also what makes them deadlock-free on hardware);
 * memory-model effects (L1/L2 visibility) are not modelled; wavefront = 64.
 */
#ifndef SIMT_EMU_HIP_RUNTIME_H
#define SIMT_EMU_HIP_RUNTIME_H

#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ucontext.h>

#include <chrono>
#include <functional>
#include <vector>

#define SIMT_EMU 1

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__ static
#define __constant__ static
#define __launch_bounds__(...)
#ifndef __restrict__
#define __restrict__
#endif
#define HIP_KERNEL_NAME(...) __VA_ARGS__

struct uint3 { unsigned x, y, z; };
struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint4 { unsigned x, y, z, w; };
struct uint2 { unsigned x, y; };
struct int2 { int x, y; };
struct int4 { int x, y, z, w; };
static inline uint4 make_uint4(unsigned a, unsigned b, unsigned c, unsigned d) { uint4 r = {a, b, c, d}; return r; }
static inline uint2 make_uint2(unsigned a, unsigned b) { uint2 r = {a, b}; return r; }
static inline int2 make_int2(int a, int b) { int2 r = {a, b}; return r; }

extern uint3 threadIdx, blockIdx;
extern dim3 blockDim, gridDim;
static const int warpSize = 64;

/* ---- error / runtime API ---- */
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorNoDevice = 100, hipErrorNotReady = 600, hipErrorLaunchFailure = 719 };
typedef struct simt_stream* hipStream_t;
typedef struct simt_event* hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
enum { hipHostMallocDefault = 0, hipHostMallocMapped = 2 };
struct hipDeviceProp_t { char name[256]; int multiProcessorCount; char gcnArchName[256]; size_t totalGlobalMem; };

const char* hipGetErrorString(hipError_t e);
hipError_t hipGetLastError(void);
hipError_t hipGetDeviceCount(int* n);
hipError_t hipSetDevice(int d);
hipError_t hipGetDevice(int* d);
hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int d);
hipError_t hipMalloc(void** p, size_t n);
template <class T> static inline hipError_t hipMalloc(T** p, size_t n) { return hipMalloc((void**)p, n); }
hipError_t hipFree(void* p);
hipError_t hipHostMalloc(void** p, size_t n, unsigned flags = 0);
template <class T> static inline hipError_t hipHostMalloc(T** p, size_t n, unsigned flags = 0) { return hipHostMalloc((void**)p, n, flags); }
hipError_t hipHostFree(void* p);
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind k);
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind k, hipStream_t st = 0);
hipError_t hipMemcpy2D(void* d, size_t dp, const void* s, size_t sp, size_t w, size_t h, hipMemcpyKind k);
hipError_t hipMemcpy2DAsync(void* d, size_t dp, const void* s, size_t sp, size_t w, size_t h, hipMemcpyKind k, hipStream_t st = 0);
hipError_t hipMemcpyPeerAsync(void* d, int ddev, const void* s, int sdev, size_t n, hipStream_t st = 0);
hipError_t hipMemset(void* d, int v, size_t n);
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t st = 0);
hipError_t hipStreamCreate(hipStream_t* s);
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned flags);
hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned flags, int priority);
hipError_t hipDeviceGetStreamPriorityRange(int* least, int* greatest);
hipError_t hipStreamDestroy(hipStream_t s);
hipError_t hipStreamSynchronize(hipStream_t s);
hipError_t hipDeviceSynchronize(void);
hipError_t hipEventCreate(hipEvent_t* e);
enum { hipEventDefault = 0, hipEventDisableTiming = 2 };
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
/* interprocess memory / events (runtime_ipc.hip): the CPU tier runs the ranks as THREADS of one process — a "handle" is the pointer itself, an
   opened event a fresh (inert) one: every launch is synchronous here, what the tier checks is the transport's bookkeeping (segment, sequence words,
   export table, per-picture meeting), not the ordering on a device */
enum { hipEventInterprocess = 4, hipIpcMemLazyEnablePeerAccess = 1 };
struct hipIpcMemHandle_t { char reserved[64]; };
struct hipIpcEventHandle_t { char reserved[64]; };
static inline hipError_t hipIpcGetMemHandle(hipIpcMemHandle_t* h, void* p) { memset(h, 0, sizeof(*h)); memcpy(h->reserved, &p, sizeof(p)); return hipSuccess; }
static inline hipError_t hipIpcOpenMemHandle(void** p, hipIpcMemHandle_t h, unsigned) { memcpy(p, h.reserved, sizeof(*p)); return hipSuccess; }
static inline hipError_t hipIpcCloseMemHandle(void*) { return hipSuccess; }
static inline hipError_t hipIpcGetEventHandle(hipIpcEventHandle_t* h, hipEvent_t e) { memset(h, 0, sizeof(*h)); memcpy(h->reserved, &e, sizeof(e)); return hipSuccess; }
static inline hipError_t hipIpcOpenEventHandle(hipEvent_t* e, hipIpcEventHandle_t) { return hipEventCreate(e); }
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s = 0);
hipError_t hipEventSynchronize(hipEvent_t e);
static inline hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned flags = 0);
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b);
enum { hipStreamNonBlocking = 1 };

/* ---- the interpreter ---- */
namespace simt {
void launch(dim3 grid, dim3 block, const std::function<void()>& body, const char* kernel_name = nullptr);
void sync_threads();
void wave_rendezvous();                 /* all live lanes of the calling lane's wave */
unsigned long long* wave_slots();       /* 64 exchange slots of the calling lane's wave */
int lane_id();
unsigned long long live_mask();         /* live (not yet returned) lanes of the wave */
}

template <class... KArgs, class... Args>
static inline void simt_launch_named(const char* name, void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t shmem, hipStream_t stream, Args... args)
{
  (void)shmem; (void)stream;
  simt::launch(grid, block, [=]() { kernel(args...); }, name);
}
/* (a macro, as in HIP: the kernel's spelling reaches the interpreter, which orders blocks per kernel — simt_emu.cpp) */
#define hipLaunchKernelGGL(kernel, ...) simt_launch_named(#kernel, kernel, __VA_ARGS__)

static inline void __syncthreads() { simt::sync_threads(); }
static inline void __builtin_amdgcn_wave_barrier() { simt::wave_rendezvous(); }
static inline void __builtin_amdgcn_s_barrier() { simt::sync_threads(); }
#define __builtin_amdgcn_fence(order, scope) ((void)0)
#define __builtin_amdgcn_sched_barrier(mask) ((void)0)
static inline void __builtin_amdgcn_s_sleep(int) {}
/* v_alignbit_b32 / v_alignbyte_b32: low 32 bits of {hi:lo} >> shift */
static inline unsigned __builtin_amdgcn_alignbit(unsigned hi, unsigned lo, unsigned s) { return (unsigned)((((unsigned long long)hi << 32) | lo) >> (s & 31)); }
static inline unsigned __builtin_amdgcn_alignbyte(unsigned hi, unsigned lo, unsigned s) { return (unsigned)((((unsigned long long)hi << 32) | lo) >> ((s & 3) * 8)); }
static inline void __threadfence() {}
static inline void __threadfence_block() {}
static inline void __threadfence_system() {}
static inline unsigned long long wall_clock64() { return 0; }      /* (only the interprocess transport's wait kernel reads it, and the interpreter never launches that) */

template <class T> static inline T simt_exchange(T v, int src)
{
  static_assert(sizeof(T) <= 8, "shuffle of >8 byte type");
  unsigned long long* s = simt::wave_slots();
  unsigned long long raw = 0;
  memcpy(&raw, &v, sizeof(T));
  const int lane = simt::lane_id();
  s[lane] = raw;
  simt::wave_rendezvous();
  unsigned long long got = ((simt::live_mask() >> (src & 63)) & 1) ? s[src & 63] : raw;
  simt::wave_rendezvous();
  T r; memcpy(&r, &got, sizeof(T));
  return r;
}
template <class T> static inline T __shfl(T v, int src, int width = 64)
{
  const int lane = simt::lane_id();
  const int base = lane & ~(width - 1);
  return simt_exchange(v, base + (src & (width - 1)));
}
template <class T> static inline T __shfl_xor(T v, int mask, int width = 64)
{
  const int lane = simt::lane_id();
  int src = lane ^ mask;
  if ((src & ~(width - 1)) != (lane & ~(width - 1))) src = lane;
  return simt_exchange(v, src);
}
template <class T> static inline T __shfl_up(T v, unsigned d, int width = 64)
{
  const int lane = simt::lane_id();
  int src = lane - (int)d;
  if (src < (lane & ~(width - 1))) src = lane;
  return simt_exchange(v, src);
}
template <class T> static inline T __shfl_down(T v, unsigned d, int width = 64)
{
  const int lane = simt::lane_id();
  int src = lane + (int)d;
  if (src > (lane | (width - 1))) src = lane;
  return simt_exchange(v, src);
}
static inline unsigned long long __ballot(int pred)
{
  unsigned long long* s = simt::wave_slots();
  s[simt::lane_id()] = pred ? 1 : 0;
  simt::wave_rendezvous();
  unsigned long long m = 0, live = simt::live_mask();
  for (int i = 0; i < 64; i++)
    if (((live >> i) & 1) && s[i]) m |= 1ull << i;
  simt::wave_rendezvous();
  return m;
}
static inline int __any(int pred) { return __ballot(pred) != 0; }
static inline int __all(int pred) { return __ballot(!pred) == 0; }
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __popc(unsigned v) { return __builtin_popcount(v); }
static inline int __ffsll(unsigned long long v) { return __builtin_ffsll((long long)v); }
static inline int __ffs(unsigned v) { return __builtin_ffs((int)v); }
static inline int __clzll(unsigned long long v) { return v ? __builtin_clzll(v) : 64; }
static inline int __clz(unsigned v) { return v ? __builtin_clz(v) : 32; }
template <class T> static inline T __builtin_amdgcn_readfirstlane(T v)
{
  const unsigned long long live = simt::live_mask();
  return simt_exchange(v, __builtin_ctzll(live));
}

template <class T> static inline T __builtin_amdgcn_readlane(T v, int lane) { return simt_exchange(v, lane & 63); }

/* ds_bpermute_b32: lane i reads `v` of lane (addr / 4) & 63 */
static inline int __builtin_amdgcn_ds_bpermute(int addr, int v) { return simt_exchange(v, (addr >> 2) & 63); }
/* v_mov_b32 dpp (all lanes of the wave active, row / bank masks 0xF): the controls the kernels use */
static inline int __builtin_amdgcn_update_dpp(int old, int v, int ctrl, int row_mask, int bank_mask, bool bound_ctrl)
{
  (void)row_mask; (void)bank_mask; (void)bound_ctrl;
  const int lane = simt::lane_id();
  int src = lane;
  bool valid = true;
  if (ctrl < 0x100) src = (lane & ~3) | ((ctrl >> (2 * (lane & 3))) & 3);        /* quad_perm */
  else if (ctrl == 0x130) { src = lane + 1; valid = src < 64; }                    /* wave_shl:1 */
  else if (ctrl == 0x138) { src = lane - 1; valid = src >= 0; }                    /* wave_shr:1 */
  else if (ctrl == 0x140) src = (lane & ~15) | (15 - (lane & 15));                 /* row_mirror */
  else if (ctrl == 0x141) src = (lane & ~7) | (7 - (lane & 7));                    /* row_half_mirror */
  else abort();
  const int got = simt_exchange(v, valid ? src : lane);
  return valid ? got : old;
}

/* atomics: one OS thread, fibers switch only at rendezvous points */
template <class T> static inline T atomicAdd(T* p, T v) { T o = *p; *p = o + v; return o; }
template <class T> static inline T atomicOr(T* p, T v) { T o = *p; *p = o | v; return o; }
template <class T> static inline T atomicXor(T* p, T v) { T o = *p; *p = o ^ v; return o; }
template <class T> static inline T atomicMax(T* p, T v) { T o = *p; if (v > o) *p = v; return o; }
template <class T> static inline T atomicMin(T* p, T v) { T o = *p; if (v < o) *p = v; return o; }
template <class T> static inline T atomicExch(T* p, T v) { T o = *p; *p = v; return o; }
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __HIP_MEMORY_SCOPE_WORKGROUP 2
#define __hip_atomic_load(p, order, scope) (*(p))
#define __hip_atomic_store(p, v, order, scope) (*(p) = (v))
#define __hip_atomic_fetch_add(p, v, order, scope) atomicAdd((p), (v))

static inline int min(int a, int b) { return a < b ? a : b; }
static inline int max(int a, int b) { return a > b ? a : b; }
static inline unsigned min(unsigned a, unsigned b) { return a < b ? a : b; }
static inline unsigned max(unsigned a, unsigned b) { return a > b ? a : b; }
static inline int __mul24(int a, int b) { return a * b; }
static inline int __builtin_amdgcn_sdot2(int a, int b, int c, bool) {
  return c + (int)(int16_t)(a & 0xFFFF) * (int16_t)(b & 0xFFFF) + (int)(int16_t)((unsigned)a >> 16) * (int16_t)((unsigned)b >> 16);
}

#endif
