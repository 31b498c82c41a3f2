/*
 * slots.hip — the SLOT LAYER: init_acceleration_functions_mi355x() fills a table with the exact
 * member order and signatures of the reference's `struct acceleration_functions`
 * (libde265/acceleration.h:29-231) — the reference's own plugin interface for this path, cf.
 * init_acceleration_functions_fallback (fallback.cc:28) / _sse (x86/sse.cc:46), call site
 * base_context::set_acceleration_functions (decctx.cc:239-270).
 *
 * Every slot is synchronous and takes HOST pointers, like the reference's: the block is staged
 * through a per-thread pinned arena (slots are called concurrently from up to 32 pool threads,
 * threads.h:86, so all state is thread_local), the HIP kernel for that one block runs on the
 * thread's own stream, and the result is copied back.  One PCIe round trip per block: this layer is
 * the parity/compatibility entry (driven by tests like dev-tools/test-*.cc drive the SIMD tables),
 * the picture layer (runtime.hip) is the fast path.  There is no CPU arithmetic here: if HIP fails
 * the slot aborts loudly (the reference's slots have no error channel, acceleration.h).
 *
 * Arithmetic follows fallback-motion.cc:33-636, fallback-dct.cc:81-859/1212-1220,
 * fallback-deblk.h:33-124 and intrapred.h:261-433 exactly.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "k_common.h"

/* ------------------------------------------------------------------------------ kernels ------- */

__constant__ int8_t cs_qpel[4][8] = {{0, 0, 0, 64, 0, 0, 0, 0}, {-1, 4, -10, 58, 17, -5, 1, 0}, {-1, 4, -11, 40, 40, -11, 4, -1}, {0, 1, -5, 17, 58, -10, 4, -1}};
__constant__ int8_t cs_epel[8][8] = {{0, 0, 0, 64, 0, 0, 0, 0},   {0, 0, -2, 58, 10, -2, 0, 0}, {0, 0, -4, 54, 16, -2, 0, 0}, {0, 0, -6, 46, 28, -4, 0, 0},
                                     {0, 0, -4, 36, 36, -4, 0, 0}, {0, 0, -4, 28, 46, -6, 0, 0}, {0, 0, -2, 16, 54, -4, 0, 0}, {0, 0, -2, 10, 58, -2, 0, 0}};
__constant__ int8_t cs_dct_qw[33] = {64, 90, 90, 90, 89, 88, 87, 85, 83, 82, 80, 78, 75, 73, 70, 67, 64,
                                     61, 57, 54, 50, 46, 43, 38, 36, 31, 25, 22, 18, 13, 9,  4,  0};
__constant__ int8_t cs_dst4[16] = {29, 55, 74, 84, 74, 74, 0, -74, 84, -29, -74, 55, 55, -84, 74, -29};
__constant__ int8_t cs_angle[35] = {0,   0,   32,  26,  21,  17, 13, 9,  5, 2, 0, -2, -5, -9, -13, -17, -21, -26,
                                    -32, -26, -21, -17, -13, -9, -5, -2, 0, 2, 5, 9,  13, 17, 21,  26,  32};
__constant__ int16_t cs_inv_angle[15] = {-4096, -1638, -910, -630, -482, -390, -315, -256, -315, -390, -482, -630, -910, -1638, -4096};

__device__ __forceinline__ int s_dct(int m)
{
  m &= 127;
  if (m <= 32) return cs_dct_qw[m];
  if (m <= 64) return -cs_dct_qw[64 - m];
  if (m < 96) return -cs_dct_qw[m - 64];
  return cs_dct_qw[128 - m];
}

/* put_hevc_qpel / put_hevc_epel* (fallback-motion.cc:492-636, 305-415, 431-485, 262-302).
 * win: packed (w+7) x (h+7) window whose sample (3,3) is the block origin; out: packed w x h. */
template <class PIX>
__global__ void __launch_bounds__(256) ks_mc(int16_t* out, const PIX* win, int w, int h, int xf, int yf, int chroma, int bd)
{
  __shared__ int16_t s_tmp[71 * 64];
  const int pitch = w + 7, shift1 = bd - 8, shift3 = max(2, 14 - bd);
  const int8_t* tx = chroma ? cs_epel[xf] : cs_qpel[xf];
  const int8_t* ty = chroma ? cs_epel[yf] : cs_qpel[yf];
  for (int i = threadIdx.x; i < (h + 7) * w; i += 256) {
    const int r = i / w, x = i - r * w;
    int v;
    if (xf == 0) v = win[r * pitch + x + 3];
    else {
      int s = 0;
      for (int k = 0; k < 8; k++) s += tx[k] * (int)win[r * pitch + x + k];
      v = s >> shift1;
    }
    s_tmp[r * 64 + x] = (int16_t)v;
  }
  __syncthreads();
  const int vshift = xf == 0 ? shift1 : 6;
  for (int i = threadIdx.x; i < h * w; i += 256) {
    const int y = i / w, x = i - y * w;
    int v;
    if (yf == 0) {
      v = s_tmp[(y + 3) * 64 + x];
      if (xf == 0) v = (int)((unsigned)v & 0xFFFF) << shift3;
    } else {
      int s = 0;
      for (int t = 0; t < 8; t++) s += ty[t] * (int)s_tmp[(y + t) * 64 + x];
      v = s >> vshift;
    }
    out[i] = (int16_t)v;
  }
}

/* put_unweighted_pred / put_weighted_pred_avg / put_weighted_pred / put_weighted_bipred
 * (fallback-motion.cc:33-256); mode 0..3 in that order; all buffers packed w x h */
template <class PIX>
__global__ void __launch_bounds__(256) ks_wpred(PIX* dst, const int16_t* s1, const int16_t* s2, int n, int mode, int w1, int o1, int w2, int o2, int log2WD, int bd)
{
  const int shift3 = max(2, 14 - bd);
  for (int i = threadIdx.x + blockIdx.x * 256; i < n; i += 256 * gridDim.x) {
    int v;
    if (mode == 0) v = (s1[i] + (1 << (shift3 - 1))) >> shift3;
    else if (mode == 1) { const int sh = max(3, 15 - bd); v = (s1[i] + s2[i] + (1 << (sh - 1))) >> sh; }
    else if (mode == 2) v = ((s1[i] * w1 + (1 << (log2WD - 1))) >> log2WD) + o1;
    else v = (s1[i] * w1 + s2[i] * w2 + (int)((unsigned)(o1 + o2 + 1) << log2WD)) >> (log2WD + 1);
    dst[i] = (PIX)d_clip_bd(v, bd);
  }
}

/* transform_idct_fallback / transform_idst_4x4_fallback (fallback-dct.cc:469-508, 695-859) and, with
 * dst != nullptr, transform_idct_add / transform_4x4_luma_add (:550-691, 269-407). One block. */
template <class PIX>
__global__ void __launch_bounds__(256) ks_transform(int32_t* res, PIX* dst, const int16_t* coeffs, int log2, int is_dst, int bdShift, int max_coeff_bits, int bd)
{
  __shared__ int16_t s_g[1024];
  const int nT = 1 << log2, fact = 32 >> log2;
  const int cmax = (1 << max_coeff_bits) - 1, cmin = -(1 << max_coeff_bits);
  for (int o = threadIdx.x; o < nT * nT; o += 256) {
    const int c = o & (nT - 1), i = o >> log2;
    int sum = 0;
    for (int j = 0; j < nT; j++) sum += (is_dst ? cs_dst4[j * 4 + i] : s_dct((fact * j) * (2 * i + 1))) * (int)coeffs[c + j * nT];
    s_g[c + i * nT] = (int16_t)d_clip3(cmin, cmax, (sum + 64) >> 7);
  }
  __syncthreads();
  const int rnd2 = 1 << (bdShift - 1);
  for (int o = threadIdx.x; o < nT * nT; o += 256) {
    const int i = o & (nT - 1), y = o >> log2;
    int sum = 0;
    for (int j = 0; j < nT; j++) sum += (is_dst ? cs_dst4[j * 4 + i] : s_dct((fact * j) * (2 * i + 1))) * (int)s_g[y * nT + j];
    const int r = (sum + rnd2) >> bdShift;
    if (dst) dst[o] = (PIX)d_clip_bd((int)dst[o] + r, bd);
    else res[o] = r;
  }
}

/* transform_skip_residual / rdpcm_v / rdpcm_h / transform_bypass[_rdpcm_v/_h]
 * (fallback-dct.cc:81-91, 161-256) -> int32 residual, or (dst != nullptr) added to dst as
 * transform_skip_rdpcm_v/h_8 do (:94-134). skip: apply tsShift/bdShift; dir 0 none, 1 vertical, 2 horizontal */
template <class PIX>
__global__ void __launch_bounds__(64) ks_resid_misc(int32_t* res, PIX* dst, const int16_t* coeffs, int nT, int skip, int dir, int tsShift, int bdShift, int bd)
{
  const int rnd = skip ? (1 << (bdShift - 1)) : 0;
  const int t = threadIdx.x;
  if (dir == 0) {
    for (int o = t; o < nT * nT; o += 64) {
      int c = coeffs[o];
      if (skip) c = ((int32_t)((uint32_t)c << tsShift) + rnd) >> bdShift;
      if (dst) dst[o] = (PIX)d_clip_bd((int)dst[o] + c, bd); else res[o] = c;
    }
    return;
  }
  if (t >= nT) return;
  int sum = 0;
  for (int k = 0; k < nT; k++) {
    const int o = dir == 1 ? t + k * nT : k + t * nT;
    int c = coeffs[o];
    if (skip) c = ((int32_t)((uint32_t)c << tsShift) + rnd) >> bdShift;
    sum += c;
    if (dst) dst[o] = (PIX)d_clip_bd((int)dst[o] + sum, bd); else res[o] = sum;
  }
}

/* add_residual (fallback-dct.h:65-73) */
template <class PIX>
__global__ void __launch_bounds__(256) ks_add_residual(PIX* dst, const int32_t* r, int n, int bd)
{
  for (int i = threadIdx.x; i < n; i += 256) dst[i] = (PIX)d_clip_bd((int)dst[i] + r[i], bd);
}

/* dequant_coeff_block (fallback-dct.cc:1212-1220): scatter into the caller's (uploaded) coeffBuf */
__global__ void __launch_bounds__(256) ks_dequant(int16_t* buf, const int16_t* list, const int16_t* pos, int n, int fact, int offset, int bdShift)
{
  for (int i = threadIdx.x; i < n; i += 256) {
    const int32_t v = (list[i] * fact + offset) >> bdShift;
    buf[pos[i]] = (int16_t)d_clip3(-32768, 32767, v);
  }
}

/* rotate_coefficients (fallback-dct.cc:228-256): 180 degree rotation in place */
__global__ void __launch_bounds__(256) ks_rotate(int16_t* c, int n)
{
  for (int i = threadIdx.x; i < n / 2; i += 256) { const int16_t a = c[i], b = c[n - 1 - i]; c[i] = b; c[n - 1 - i] = a; }
}

/* deblock_luma_kernel / deblock_chroma_kernel (fallback-deblk.h:33-124) on a packed 8x4 (luma) or
 * 4x4 (chroma) patch: row k = line k along the edge, columns = p3..p0 q0..q3 (resp. p1 p0 q0 q1) */
template <class PIX>
__global__ void __launch_bounds__(64) ks_deblock(PIX* patch, int luma, int dE, int dEp, int dEq, int tc, int filterP, int filterQ, int bd)
{
  const int k = threadIdx.x;
  if (k >= 4) return;
  if (luma) {
    PIX* o = patch + k * 8 + 4;
    const int p0 = o[-1], p1 = o[-2], p2 = o[-3], p3 = o[-4], q0 = o[0], q1 = o[1], q2 = o[2], q3 = o[3];
    if (dE == 2) {
      if (filterP) {
        o[-1] = (PIX)d_clip3(p0 - 2 * tc, p0 + 2 * tc, (p2 + 2 * p1 + 2 * p0 + 2 * q0 + q1 + 4) >> 3);
        o[-2] = (PIX)d_clip3(p1 - 2 * tc, p1 + 2 * tc, (p2 + p1 + p0 + q0 + 2) >> 2);
        o[-3] = (PIX)d_clip3(p2 - 2 * tc, p2 + 2 * tc, (2 * p3 + 3 * p2 + p1 + p0 + q0 + 4) >> 3);
      }
      if (filterQ) {
        o[0] = (PIX)d_clip3(q0 - 2 * tc, q0 + 2 * tc, (p1 + 2 * p0 + 2 * q0 + 2 * q1 + q2 + 4) >> 3);
        o[1] = (PIX)d_clip3(q1 - 2 * tc, q1 + 2 * tc, (p0 + q0 + q1 + q2 + 2) >> 2);
        o[2] = (PIX)d_clip3(q2 - 2 * tc, q2 + 2 * tc, (p0 + q0 + q1 + 3 * q2 + 2 * q3 + 4) >> 3);
      }
    } else {
      int delta = (9 * (q0 - p0) - 3 * (q1 - p1) + 8) >> 4;
      if (d_abs(delta) < tc * 10) {
        delta = d_clip3(-tc, tc, delta);
        if (filterP) o[-1] = (PIX)d_clip_bd(p0 + delta, bd);
        if (filterQ) o[0] = (PIX)d_clip_bd(q0 - delta, bd);
        if (dEp == 1 && filterP) o[-2] = (PIX)d_clip_bd(p1 + d_clip3(-(tc >> 1), tc >> 1, (((p2 + p0 + 1) >> 1) - p1 + delta) >> 1), bd);
        if (dEq == 1 && filterQ) o[1] = (PIX)d_clip_bd(q1 + d_clip3(-(tc >> 1), tc >> 1, (((q2 + q0 + 1) >> 1) - q1 - delta) >> 1), bd);
      }
    }
  } else {
    PIX* o = patch + k * 4 + 2;
    const int p0 = o[-1], p1 = o[-2], q0 = o[0], q1 = o[1];
    const int delta = d_clip3(-tc, tc, ((((q0 - p0) * 4) + p1 - q1 + 4) >> 3));
    if (filterP) o[-1] = (PIX)d_clip_bd(p0 + delta, bd);
    if (filterQ) o[0] = (PIX)d_clip_bd(q0 - delta, bd);
  }
}

/* intra_prediction_planar / _DC / _angular (intrapred.h:261-433); border: packed p[-2nT..2nT], entry
 * i + 2nT; which 0 planar, 1 DC, 2 angular; out packed nT x nT */
template <class PIX>
__global__ void __launch_bounds__(256) ks_intra(PIX* out, const PIX* border, int log2, int cIdx, int which, int mode, int disableBoundaryFilter, int bd)
{
  __shared__ int s_ref[3 * 32 + 8];
  __shared__ int s_dc;
  const int nT = 1 << log2, Z = 2 * nT;
  int* ref = s_ref + 32;
#define BRD(i) ((int)border[(i) + Z])
  const int angle = which == 2 ? cs_angle[mode] : 0;
  if (which == 1 && threadIdx.x == 0) {
    int s = 0;
    for (int i = 0; i < nT; i++) s += BRD(i + 1) + BRD(-i - 1);
    s_dc = (s + nT) >> (log2 + 1);
  }
  if (which == 2) {
    const int sgn = mode >= 18 ? 1 : -1;
    const int inv = angle < 0 ? cs_inv_angle[mode - 11] : 0;
    const int lo = (nT * angle) >> 5;
    for (int t = threadIdx.x; t < 3 * nT + 1; t += 256) {
      const int x = t - nT;
      int v = 0;
      if (x >= 0 && x <= nT) v = BRD(sgn * x);
      else if (x < 0) { if (angle < 0 && lo < -1 && x >= lo) v = BRD(-sgn * ((x * inv + 128) >> 8)); }
      else if (angle >= 0) v = BRD(sgn * x);
      ref[x] = v;
    }
  }
  __syncthreads();
  const bool edge = (cIdx == 0 && nT < 32);
  for (int o = threadIdx.x; o < nT * nT; o += 256) {
    const int y = o >> log2, x = o & (nT - 1);
    int v;
    if (which == 0) {
      v = ((nT - 1 - x) * BRD(-1 - y) + (x + 1) * BRD(1 + nT) + (nT - 1 - y) * BRD(1 + x) + (y + 1) * BRD(-1 - nT) + nT) >> (log2 + 1);
    } else if (which == 1) {
      const int dc = s_dc;
      v = dc;
      if (edge) {
        if (x == 0 && y == 0) v = (BRD(-1) + 2 * dc + BRD(1) + 2) >> 2;
        else if (y == 0) v = (BRD(x + 1) + 3 * dc + 2) >> 2;
        else if (x == 0) v = (BRD(-y - 1) + 3 * dc + 2) >> 2;
      }
    } else {
      const int a = mode >= 18 ? y : x, b = mode >= 18 ? x : y;
      const int iIdx = ((a + 1) * angle) >> 5, iFact = ((a + 1) * angle) & 31;
      v = iFact ? ((32 - iFact) * ref[b + iIdx + 1] + iFact * ref[b + iIdx + 2] + 16) >> 5 : ref[b + iIdx + 1];
      if (edge && !disableBoundaryFilter) {
        if (mode == 26 && x == 0) v = d_clip_bd(BRD(1) + ((BRD(-1 - y) - BRD(0)) >> 1), bd);
        if (mode == 10 && y == 0) v = d_clip_bd(BRD(-1) + ((BRD(1 + x) - BRD(0)) >> 1), bd);
      }
    }
    out[o] = (PIX)v;
  }
#undef BRD
}

/* --------------------------------------------------------------------- per-thread staging ----- */

namespace {

constexpr size_t SEG = 64 * 1024;   /* >= (64+7)^2 * 2 B window, 64*64 int16, 32*32 int32 */
constexpr int NSEG = 4;

struct SlotTLS {
  bool ready = false;
  hipStream_t st = nullptr;
  char* dev = nullptr;
  char* host = nullptr;
  ~SlotTLS()
  {
    if (!ready) return;
    hipStreamSynchronize(st);
    hipFree(dev); hipHostFree(host); hipStreamDestroy(st);
  }
};
thread_local SlotTLS g_tls;
int g_slot_device = 0;

[[noreturn]] void slot_die(const char* what, hipError_t e)
{
  fprintf(stderr, "libde265_mi355x: slot layer: %s failed: %s (the MI355X backend has no CPU fallback)\n", what, hipGetErrorString(e));
  abort();
}
#define SCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) slot_die(#call, e_); } while (0)

SlotTLS& tls()
{
  SlotTLS& t = g_tls;
  if (!t.ready) {
    SCHK(hipSetDevice(g_slot_device));
    SCHK(hipStreamCreateWithFlags(&t.st, hipStreamNonBlocking));
    SCHK(hipMalloc(&t.dev, SEG * NSEG));
    SCHK(hipHostMalloc(&t.host, SEG * NSEG, hipHostMallocDefault));
    t.ready = true;
  }
  return t;
}
template <class T> T* hseg(SlotTLS& t, int i) { return (T*)(t.host + SEG * i); }
template <class T> T* dseg(SlotTLS& t, int i) { return (T*)(t.dev + SEG * i); }
void up(SlotTLS& t, int i, size_t bytes) { SCHK(hipMemcpyAsync(t.dev + SEG * i, t.host + SEG * i, bytes, hipMemcpyHostToDevice, t.st)); }
void down(SlotTLS& t, int i, size_t bytes) { SCHK(hipMemcpyAsync(t.host + SEG * i, t.dev + SEG * i, bytes, hipMemcpyDeviceToHost, t.st)); }
void sync(SlotTLS& t)
{
  SCHK(hipGetLastError());
  SCHK(hipStreamSynchronize(t.st));
}
template <class T> void pack(T* dst, const T* src, ptrdiff_t stride, int w, int h)
{
  for (int y = 0; y < h; y++) memcpy(dst + (size_t)y * w, src + y * stride, sizeof(T) * w);
}
template <class T> void unpack(T* dst, ptrdiff_t stride, const T* src, int w, int h)
{
  for (int y = 0; y < h; y++) memcpy(dst + y * stride, src + (size_t)y * w, sizeof(T) * w);
}

/* ---- weighted prediction ---- */
template <class PIX>
void wpred(PIX* dst, ptrdiff_t ds, const int16_t* s1, const int16_t* s2, ptrdiff_t ss, int w, int h, int mode, int w1, int o1, int w2, int o2, int log2WD, int bd)
{
  SlotTLS& t = tls();
  const int n = w * h;
  pack(hseg<int16_t>(t, 1), s1, ss, w, h);
  up(t, 1, 2 * (size_t)n);
  if (s2) { pack(hseg<int16_t>(t, 2), s2, ss, w, h); up(t, 2, 2 * (size_t)n); }
  hipLaunchKernelGGL(HIP_KERNEL_NAME(ks_wpred<PIX>), dim3((n + 255) / 256), dim3(256), 0, t.st, dseg<PIX>(t, 0), dseg<int16_t>(t, 1), dseg<int16_t>(t, 2), n, mode, w1, o1, w2, o2, log2WD, bd);
  down(t, 0, sizeof(PIX) * (size_t)n);
  sync(t);
  unpack(dst, ds, hseg<PIX>(t, 0), w, h);
}

/* ---- interpolation ---- */
const int kExtraBefore[4] = {0, 3, 3, 2}, kExtraAfter[4] = {0, 3, 4, 4};   /* fallback-motion.cc:489-490 */
template <class PIX>
void mc(int16_t* out, ptrdiff_t os, const PIX* src, ptrdiff_t ss, int w, int h, int xf, int yf, int chroma, int bd)
{
  SlotTLS& t = tls();
  const int pitch = w + 7;
  PIX* win = hseg<PIX>(t, 1);
  memset(win, 0, sizeof(PIX) * (size_t)pitch * (h + 7));
  /* copy exactly the rectangle the reference's slot reads (its caller guarantees nothing more) */
  const int bx = chroma ? (xf ? 1 : 0) : kExtraBefore[xf], ax = chroma ? (xf ? 2 : 0) : kExtraAfter[xf];
  const int by = chroma ? (yf ? 1 : 0) : kExtraBefore[yf], ay = chroma ? (yf ? 2 : 0) : kExtraAfter[yf];
  for (int y = -by; y < h + ay; y++) memcpy(win + (size_t)(y + 3) * pitch + 3 - bx, src + y * ss - bx, sizeof(PIX) * (w + bx + ax));
  up(t, 1, sizeof(PIX) * (size_t)pitch * (h + 7));
  hipLaunchKernelGGL(HIP_KERNEL_NAME(ks_mc<PIX>), dim3(1), dim3(256), 0, t.st, dseg<int16_t>(t, 0), dseg<PIX>(t, 1), w, h, xf, yf, chroma, bd);
  down(t, 0, 2 * (size_t)w * h);
  sync(t);
  unpack(out, os, hseg<int16_t>(t, 0), w, h);
}

/* ---- transforms ---- */
template <class PIX>
void transform_add(PIX* dst, const int16_t* coeffs, ptrdiff_t stride, int log2, int is_dst, int bd)
{
  SlotTLS& t = tls();
  const int nT = 1 << log2, n = nT * nT;
  memcpy(hseg<int16_t>(t, 1), coeffs, 2 * (size_t)n);
  pack(hseg<PIX>(t, 0), dst, stride, nT, nT);
  up(t, 1, 2 * (size_t)n); up(t, 0, sizeof(PIX) * (size_t)n);
  hipLaunchKernelGGL(HIP_KERNEL_NAME(ks_transform<PIX>), dim3(1), dim3(256), 0, t.st, (int32_t*)nullptr, dseg<PIX>(t, 0), dseg<int16_t>(t, 1), log2, is_dst, 20 - bd, 15, bd);
  down(t, 0, sizeof(PIX) * (size_t)n);
  sync(t);
  unpack(dst, stride, hseg<PIX>(t, 0), nT, nT);
}
void transform_res(int32_t* dst, const int16_t* coeffs, int log2, int is_dst, int bdShift, int max_coeff_bits)
{
  SlotTLS& t = tls();
  const int n = 1 << (2 * log2);
  memcpy(hseg<int16_t>(t, 1), coeffs, 2 * (size_t)n);
  up(t, 1, 2 * (size_t)n);
  hipLaunchKernelGGL(HIP_KERNEL_NAME(ks_transform<uint8_t>), dim3(1), dim3(256), 0, t.st, dseg<int32_t>(t, 0), (uint8_t*)nullptr, dseg<int16_t>(t, 1), log2, is_dst, bdShift, max_coeff_bits, 8);
  down(t, 0, 4 * (size_t)n);
  sync(t);
  memcpy(dst, hseg<int32_t>(t, 0), 4 * (size_t)n);
}
void resid_misc(int32_t* r, const int16_t* coeffs, int nT, int skip, int dir, int tsShift, int bdShift)
{
  SlotTLS& t = tls();
  const int n = nT * nT;
  memcpy(hseg<int16_t>(t, 1), coeffs, 2 * (size_t)n);
  up(t, 1, 2 * (size_t)n);
  hipLaunchKernelGGL(HIP_KERNEL_NAME(ks_resid_misc<uint8_t>), dim3(1), dim3(64), 0, t.st, dseg<int32_t>(t, 0), (uint8_t*)nullptr, dseg<int16_t>(t, 1), nT, skip, dir, tsShift, bdShift, 8);
  down(t, 0, 4 * (size_t)n);
  sync(t);
  memcpy(r, hseg<int32_t>(t, 0), 4 * (size_t)n);
}
void skip_rdpcm_add_8(uint8_t* dst, const int16_t* coeffs, int log2nT, ptrdiff_t stride, int dir)
{
  SlotTLS& t = tls();
  const int nT = 1 << log2nT, n = nT * nT;
  memcpy(hseg<int16_t>(t, 1), coeffs, 2 * (size_t)n);
  pack(hseg<uint8_t>(t, 0), dst, stride, nT, nT);
  up(t, 1, 2 * (size_t)n); up(t, 0, (size_t)n);
  hipLaunchKernelGGL(HIP_KERNEL_NAME(ks_resid_misc<uint8_t>), dim3(1), dim3(64), 0, t.st, (int32_t*)nullptr, dseg<uint8_t>(t, 0), dseg<int16_t>(t, 1), nT, 1, dir, 5 + log2nT, 12, 8);
  down(t, 0, (size_t)n);
  sync(t);
  unpack(dst, stride, hseg<uint8_t>(t, 0), nT, nT);
}
template <class PIX>
void add_residual(PIX* dst, ptrdiff_t stride, const int32_t* r, int nT, int bd)
{
  SlotTLS& t = tls();
  const int n = nT * nT;
  memcpy(hseg<int32_t>(t, 1), r, 4 * (size_t)n);
  pack(hseg<PIX>(t, 0), dst, stride, nT, nT);
  up(t, 1, 4 * (size_t)n); up(t, 0, sizeof(PIX) * (size_t)n);
  hipLaunchKernelGGL(HIP_KERNEL_NAME(ks_add_residual<PIX>), dim3(1), dim3(256), 0, t.st, dseg<PIX>(t, 0), dseg<int32_t>(t, 1), n, bd);
  down(t, 0, sizeof(PIX) * (size_t)n);
  sync(t);
  unpack(dst, stride, hseg<PIX>(t, 0), nT, nT);
}

/* ---- deblocking ---- */
template <class PIX>
void deblock(PIX* ptr, ptrdiff_t stride, int vertical, int luma, int dE, int dEp, int dEq, int tc, int filterP, int filterQ, int bd)
{
  SlotTLS& t = tls();
  const int half = luma ? 4 : 2, wdt = 2 * half;
  PIX* patch = hseg<PIX>(t, 0);
  for (int k = 0; k < 4; k++)
    for (int i = -half; i < half; i++) patch[k * wdt + half + i] = vertical ? ptr[i + k * stride] : ptr[k + i * stride];
  up(t, 0, sizeof(PIX) * 4 * wdt);
  hipLaunchKernelGGL(HIP_KERNEL_NAME(ks_deblock<PIX>), dim3(1), dim3(64), 0, t.st, dseg<PIX>(t, 0), luma, dE, dEp, dEq, tc, filterP, filterQ, bd);
  down(t, 0, sizeof(PIX) * 4 * wdt);
  sync(t);
  /* the reference writes at most 3 (luma) / 1 (chroma) samples per side, and only on enabled sides:
     untouched samples come back unchanged, so writing the modifiable range back is equivalent */
  const int reach = luma ? 3 : 1;
  for (int k = 0; k < 4; k++)
    for (int i = -reach; i < reach; i++) {
      if ((i < 0 && !filterP) || (i >= 0 && !filterQ)) continue;
      if (vertical) ptr[i + k * stride] = patch[k * wdt + half + i]; else ptr[k + i * stride] = patch[k * wdt + half + i];
    }
}

/* ---- intra ---- */
template <class PIX>
void intra(PIX* dst, ptrdiff_t stride, int nT, int cIdx, const PIX* border, int which, int mode, int disableBoundaryFilter, int bd)
{
  SlotTLS& t = tls();
  int log2 = 2;
  while ((1 << log2) < nT) log2++;
  memcpy(hseg<PIX>(t, 1), border - 2 * nT, sizeof(PIX) * (4 * nT + 1));
  up(t, 1, sizeof(PIX) * (4 * nT + 1));
  hipLaunchKernelGGL(HIP_KERNEL_NAME(ks_intra<PIX>), dim3(1), dim3(256), 0, t.st, dseg<PIX>(t, 0), dseg<PIX>(t, 1), log2, cIdx, which, mode, disableBoundaryFilter, bd);
  down(t, 0, sizeof(PIX) * (size_t)nT * nT);
  sync(t);
  unpack(dst, stride, hseg<PIX>(t, 0), nT, nT);
}

/* ------------------------------------------------------------- the table's entry points ------- */

void put_weighted_pred_avg_8(uint8_t* d, ptrdiff_t ds, const int16_t* a, const int16_t* b, ptrdiff_t ss, int w, int h) { wpred(d, ds, a, b, ss, w, h, 1, 0, 0, 0, 0, 0, 8); }
void put_unweighted_pred_8(uint8_t* d, ptrdiff_t ds, const int16_t* a, ptrdiff_t ss, int w, int h) { wpred(d, ds, a, (const int16_t*)nullptr, ss, w, h, 0, 0, 0, 0, 0, 0, 8); }
void put_weighted_pred_8(uint8_t* d, ptrdiff_t ds, const int16_t* a, ptrdiff_t ss, int w, int h, int wt, int o, int l) { wpred(d, ds, a, (const int16_t*)nullptr, ss, w, h, 2, wt, o, 0, 0, l, 8); }
void put_weighted_bipred_8(uint8_t* d, ptrdiff_t ds, const int16_t* a, const int16_t* b, ptrdiff_t ss, int w, int h, int w1, int o1, int w2, int o2, int l) { wpred(d, ds, a, b, ss, w, h, 3, w1, o1, w2, o2, l, 8); }
void put_weighted_pred_avg_16(uint16_t* d, ptrdiff_t ds, const int16_t* a, const int16_t* b, ptrdiff_t ss, int w, int h, int bd) { wpred(d, ds, a, b, ss, w, h, 1, 0, 0, 0, 0, 0, bd); }
void put_unweighted_pred_16(uint16_t* d, ptrdiff_t ds, const int16_t* a, ptrdiff_t ss, int w, int h, int bd) { wpred(d, ds, a, (const int16_t*)nullptr, ss, w, h, 0, 0, 0, 0, 0, 0, bd); }
void put_weighted_pred_16(uint16_t* d, ptrdiff_t ds, const int16_t* a, ptrdiff_t ss, int w, int h, int wt, int o, int l, int bd) { wpred(d, ds, a, (const int16_t*)nullptr, ss, w, h, 2, wt, o, 0, 0, l, bd); }
void put_weighted_bipred_16(uint16_t* d, ptrdiff_t ds, const int16_t* a, const int16_t* b, ptrdiff_t ss, int w, int h, int w1, int o1, int w2, int o2, int l, int bd) { wpred(d, ds, a, b, ss, w, h, 3, w1, o1, w2, o2, l, bd); }

/* the epel slot flavours differ only in which of mx/my the caller promises to be zero (motion.cc:229-279) */
void put_hevc_epel_8(int16_t* d, ptrdiff_t ds, const uint8_t* s, ptrdiff_t ss, int w, int h, int mx, int my, int16_t*) { (void)mx; (void)my; mc(d, ds, s, ss, w, h, 0, 0, 1, 8); }
void put_hevc_epel_h_8(int16_t* d, ptrdiff_t ds, const uint8_t* s, ptrdiff_t ss, int w, int h, int mx, int my, int16_t*, int bd) { (void)my; mc(d, ds, s, ss, w, h, mx, 0, 1, bd); }
void put_hevc_epel_v_8(int16_t* d, ptrdiff_t ds, const uint8_t* s, ptrdiff_t ss, int w, int h, int mx, int my, int16_t*, int bd) { (void)mx; mc(d, ds, s, ss, w, h, 0, my, 1, bd); }
void put_hevc_epel_hv_8(int16_t* d, ptrdiff_t ds, const uint8_t* s, ptrdiff_t ss, int w, int h, int mx, int my, int16_t*, int bd) { mc(d, ds, s, ss, w, h, mx, my, 1, bd); }
void put_hevc_epel_16(int16_t* d, ptrdiff_t ds, const uint16_t* s, ptrdiff_t ss, int w, int h, int mx, int my, int16_t*, int bd) { (void)mx; (void)my; mc(d, ds, s, ss, w, h, 0, 0, 1, bd); }
void put_hevc_epel_h_16(int16_t* d, ptrdiff_t ds, const uint16_t* s, ptrdiff_t ss, int w, int h, int mx, int my, int16_t*, int bd) { (void)my; mc(d, ds, s, ss, w, h, mx, 0, 1, bd); }
void put_hevc_epel_v_16(int16_t* d, ptrdiff_t ds, const uint16_t* s, ptrdiff_t ss, int w, int h, int mx, int my, int16_t*, int bd) { (void)mx; mc(d, ds, s, ss, w, h, 0, my, 1, bd); }
void put_hevc_epel_hv_16(int16_t* d, ptrdiff_t ds, const uint16_t* s, ptrdiff_t ss, int w, int h, int mx, int my, int16_t*, int bd) { mc(d, ds, s, ss, w, h, mx, my, 1, bd); }

template <int XF, int YF> void qpel_8(int16_t* d, ptrdiff_t ds, const uint8_t* s, ptrdiff_t ss, int w, int h, int16_t*) { mc(d, ds, s, ss, w, h, XF, YF, 0, 8); }
template <int XF, int YF> void qpel_16(int16_t* d, ptrdiff_t ds, const uint16_t* s, ptrdiff_t ss, int w, int h, int16_t*, int bd) { mc(d, ds, s, ss, w, h, XF, YF, 0, bd); }

void transform_bypass(int32_t* r, const int16_t* c, int nT) { resid_misc(r, c, nT, 0, 0, 0, 1); }
void transform_bypass_rdpcm_v(int32_t* r, const int16_t* c, int nT) { resid_misc(r, c, nT, 0, 1, 0, 1); }
void transform_bypass_rdpcm_h(int32_t* r, const int16_t* c, int nT) { resid_misc(r, c, nT, 0, 2, 0, 1); }
void transform_skip_rdpcm_v_8(uint8_t* d, const int16_t* c, int log2nT, ptrdiff_t st) { skip_rdpcm_add_8(d, c, log2nT, st, 1); }
void transform_skip_rdpcm_h_8(uint8_t* d, const int16_t* c, int log2nT, ptrdiff_t st) { skip_rdpcm_add_8(d, c, log2nT, st, 2); }
void transform_4x4_dst_add_8(uint8_t* d, const int16_t* c, ptrdiff_t st) { transform_add(d, c, st, 2, 1, 8); }
template <int L> void transform_add_8(uint8_t* d, const int16_t* c, ptrdiff_t st) { transform_add(d, c, st, L, 0, 8); }
void transform_4x4_dst_add_16(uint16_t* d, const int16_t* c, ptrdiff_t st, int bd) { transform_add(d, c, st, 2, 1, bd); }
template <int L> void transform_add_16(uint16_t* d, const int16_t* c, ptrdiff_t st, int bd) { transform_add(d, c, st, L, 0, bd); }
void rotate_coefficients(int16_t* c, int nT)
{
  SlotTLS& t = tls();
  const int n = nT * nT;
  memcpy(hseg<int16_t>(t, 0), c, 2 * (size_t)n);
  up(t, 0, 2 * (size_t)n);
  hipLaunchKernelGGL(ks_rotate, dim3(1), dim3(256), 0, t.st, dseg<int16_t>(t, 0), n);
  down(t, 0, 2 * (size_t)n);
  sync(t);
  memcpy(c, hseg<int16_t>(t, 0), 2 * (size_t)n);
}
void transform_idst_4x4(int32_t* d, const int16_t* c, int bdShift, int mcb) { transform_res(d, c, 2, 1, bdShift, mcb); }
template <int L> void transform_idct(int32_t* d, const int16_t* c, int bdShift, int mcb) { transform_res(d, c, L, 0, bdShift, mcb); }
void add_residual_8(uint8_t* d, ptrdiff_t st, const int32_t* r, int nT, int bd) { add_residual(d, st, r, nT, bd); }
void add_residual_16(uint16_t* d, ptrdiff_t st, const int32_t* r, int nT, int bd) { add_residual(d, st, r, nT, bd); }

void dequant_coeff_block(int16_t* buf, const int16_t* list, const int16_t* pos, int n, int32_t fact, int32_t offset, int32_t bdShift)
{
  /* coeffBuf is the caller's 32x32 scratch (decctx.h:83-86); positions are < 1024 (slice.cc:3446) */
  if (n <= 0) return;
  SlotTLS& t = tls();
  int maxpos = 0;
  for (int i = 0; i < n; i++) if (pos[i] > maxpos) maxpos = pos[i];
  const size_t span = 2 * ((size_t)maxpos + 1);
  memcpy(hseg<int16_t>(t, 0), buf, span);
  memcpy(hseg<int16_t>(t, 1), list, 2 * (size_t)n);
  memcpy(hseg<int16_t>(t, 2), pos, 2 * (size_t)n);
  up(t, 0, span); up(t, 1, 2 * (size_t)n); up(t, 2, 2 * (size_t)n);
  hipLaunchKernelGGL(ks_dequant, dim3(1), dim3(256), 0, t.st, dseg<int16_t>(t, 0), dseg<int16_t>(t, 1), dseg<int16_t>(t, 2), n, fact, offset, bdShift);
  down(t, 0, span);
  sync(t);
  memcpy(buf, hseg<int16_t>(t, 0), span);
}

void deblock_luma_8(uint8_t* p, ptrdiff_t st, int vertical, int dE, int dEp, int dEq, int tc, int fP, int fQ) { deblock(p, st, vertical, 1, dE, dEp, dEq, tc, fP, fQ, 8); }
void deblock_chroma_8(uint8_t* p, ptrdiff_t st, int vertical, int tc, int fP, int fQ) { deblock(p, st, vertical, 0, 0, 0, 0, tc, fP, fQ, 8); }

void rdpcm_v(int32_t* r, const int16_t* c, int nT, int ts, int bs) { resid_misc(r, c, nT, 1, 1, ts, bs); }
void rdpcm_h(int32_t* r, const int16_t* c, int nT, int ts, int bs) { resid_misc(r, c, nT, 1, 2, ts, bs); }
void transform_skip_residual(int32_t* r, const int16_t* c, int nT, int ts, int bs) { resid_misc(r, c, nT, 1, 0, ts, bs); }

void intra_pred_dc_8(uint8_t* d, ptrdiff_t st, int nT, int cIdx, const uint8_t* b) { intra(d, st, nT, cIdx, b, 1, 1, 0, 8); }
void intra_pred_dc_16(uint16_t* d, ptrdiff_t st, int nT, int cIdx, const uint16_t* b) { intra(d, st, nT, cIdx, b, 1, 1, 0, 16); }
void intra_pred_planar_8(uint8_t* d, ptrdiff_t st, int nT, int cIdx, const uint8_t* b) { intra(d, st, nT, cIdx, b, 0, 0, 0, 8); }
void intra_pred_planar_16(uint16_t* d, ptrdiff_t st, int nT, int cIdx, const uint16_t* b) { intra(d, st, nT, cIdx, b, 0, 0, 0, 16); }
void intra_pred_angular_8(uint8_t* d, ptrdiff_t st, int bd, int dbf, int, int, int mode, int nT, int cIdx, const uint8_t* b) { intra(d, st, nT, cIdx, b, 2, mode, dbf, bd); }
void intra_pred_angular_16(uint16_t* d, ptrdiff_t st, int bd, int dbf, int, int, int mode, int nT, int cIdx, const uint16_t* b) { intra(d, st, nT, cIdx, b, 2, mode, dbf, bd); }

} // namespace

extern "C" int init_acceleration_functions_mi355x(void* accel)
{
  int n = 0;
  if (!accel || hipGetDeviceCount(&n) != hipSuccess || n <= 0) return M355_ERR_NO_DEVICE;   /* table untouched */
  m355_acceleration_functions* a = (m355_acceleration_functions*)accel;
  a->put_weighted_pred_avg_8 = put_weighted_pred_avg_8; a->put_unweighted_pred_8 = put_unweighted_pred_8;
  a->put_weighted_pred_8 = put_weighted_pred_8; a->put_weighted_bipred_8 = put_weighted_bipred_8;
  a->put_weighted_pred_avg_16 = put_weighted_pred_avg_16; a->put_unweighted_pred_16 = put_unweighted_pred_16;
  a->put_weighted_pred_16 = put_weighted_pred_16; a->put_weighted_bipred_16 = put_weighted_bipred_16;
  a->put_hevc_epel_8 = put_hevc_epel_8; a->put_hevc_epel_h_8 = put_hevc_epel_h_8; a->put_hevc_epel_v_8 = put_hevc_epel_v_8; a->put_hevc_epel_hv_8 = put_hevc_epel_hv_8;
  a->put_hevc_epel_16 = put_hevc_epel_16; a->put_hevc_epel_h_16 = put_hevc_epel_h_16; a->put_hevc_epel_v_16 = put_hevc_epel_v_16; a->put_hevc_epel_hv_16 = put_hevc_epel_hv_16;
#define QP(x, y) a->put_hevc_qpel_8[x][y] = qpel_8<x, y>; a->put_hevc_qpel_16[x][y] = qpel_16<x, y>;
  QP(0, 0) QP(0, 1) QP(0, 2) QP(0, 3) QP(1, 0) QP(1, 1) QP(1, 2) QP(1, 3) QP(2, 0) QP(2, 1) QP(2, 2) QP(2, 3) QP(3, 0) QP(3, 1) QP(3, 2) QP(3, 3)
#undef QP
  a->transform_bypass = transform_bypass; a->transform_bypass_rdpcm_v = transform_bypass_rdpcm_v; a->transform_bypass_rdpcm_h = transform_bypass_rdpcm_h;
  /* transform_skip_8 / transform_skip_16 are dead slots in the reference (assert(0), fallback-dct.cc:45-76): left as found */
  a->transform_skip_rdpcm_v_8 = transform_skip_rdpcm_v_8; a->transform_skip_rdpcm_h_8 = transform_skip_rdpcm_h_8;
  a->transform_4x4_dst_add_8 = transform_4x4_dst_add_8;
  a->transform_add_8[0] = transform_add_8<2>; a->transform_add_8[1] = transform_add_8<3>; a->transform_add_8[2] = transform_add_8<4>; a->transform_add_8[3] = transform_add_8<5>;
  a->transform_4x4_dst_add_16 = transform_4x4_dst_add_16;
  a->transform_add_16[0] = transform_add_16<2>; a->transform_add_16[1] = transform_add_16<3>; a->transform_add_16[2] = transform_add_16<4>; a->transform_add_16[3] = transform_add_16<5>;
  a->rotate_coefficients = rotate_coefficients;
  a->transform_idst_4x4 = transform_idst_4x4;
  a->transform_idct_4x4 = transform_idct<2>; a->transform_idct_8x8 = transform_idct<3>; a->transform_idct_16x16 = transform_idct<4>; a->transform_idct_32x32 = transform_idct<5>;
  a->add_residual_8 = add_residual_8; a->add_residual_16 = add_residual_16;
  a->dequant_coeff_block = dequant_coeff_block;
  a->deblock_luma_8 = deblock_luma_8; a->deblock_chroma_8 = deblock_chroma_8;
  a->rdpcm_v = rdpcm_v; a->rdpcm_h = rdpcm_h; a->transform_skip_residual = transform_skip_residual;
  a->intra_pred_dc_8 = intra_pred_dc_8; a->intra_pred_dc_16 = intra_pred_dc_16;
  a->intra_pred_planar_8 = intra_pred_planar_8; a->intra_pred_planar_16 = intra_pred_planar_16;
  a->intra_pred_angular_8 = intra_pred_angular_8; a->intra_pred_angular_16 = intra_pred_angular_16;
  /* encoder-only forward transforms (acceleration.h:222-230): out of scope, untouched */
  return M355_OK;
}

/* N independent blocks per launch (dense coefficients, host pointers): the batched form of
 * transform_add_8/16[log2-2] / transform_4x4_dst_add_8/16.  Blocks may not overlap. */
template <class PIX>
__global__ void __launch_bounds__(256) ks_transform_batch(PIX* base, const long long* off, ptrdiff_t stride, const int16_t* coeffs, int log2, int is_dst, int bd)
{
  __shared__ int16_t s_g[1024];
  const int nT = 1 << log2, fact = 32 >> log2;
  const int16_t* cf = coeffs + (size_t)blockIdx.x * nT * nT;
  PIX* dst = base + off[blockIdx.x];
  for (int o = threadIdx.x; o < nT * nT; o += 256) {
    const int c = o & (nT - 1), i = o >> log2;
    int sum = 0;
    for (int j = 0; j < nT; j++) sum += (is_dst ? cs_dst4[j * 4 + i] : s_dct((fact * j) * (2 * i + 1))) * (int)cf[c + j * nT];
    s_g[c + i * nT] = (int16_t)d_clip3(-32768, 32767, (sum + 64) >> 7);
  }
  __syncthreads();
  const int bdShift = 20 - bd, rnd2 = 1 << (bdShift - 1);
  for (int o = threadIdx.x; o < nT * nT; o += 256) {
    const int i = o & (nT - 1), y = o >> log2;
    int sum = 0;
    for (int j = 0; j < nT; j++) sum += (is_dst ? cs_dst4[j * 4 + i] : s_dct((fact * j) * (2 * i + 1))) * (int)s_g[y * nT + j];
    PIX* q = dst + y * stride + i;
    *q = (PIX)d_clip_bd((int)*q + ((sum + rnd2) >> bdShift), bd);
  }
}

extern "C" int m355_transform_add_batch(int n, int log2_nT, int kind, int bit_depth, void* dst_base, size_t dst_bytes,
                                        const int64_t* dst_off, ptrdiff_t stride, const int16_t* coeffs)
{
  if (n <= 0) return M355_OK;
  if (log2_nT < 2 || log2_nT > 5 || (kind == 1 && log2_nT != 2) || kind < 0 || kind > 1 || bit_depth < 8 || bit_depth > 16 || !dst_base || !dst_off || !coeffs)
    return M355_ERR_INVALID;
  const int nT = 1 << log2_nT, bpp = bit_depth <= 8 ? 1 : 2;
  for (int i = 0; i < n; i++)
    if (dst_off[i] < 0 || ((size_t)dst_off[i] + (size_t)(nT - 1) * stride + nT) * bpp > dst_bytes) return M355_ERR_INVALID;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return M355_ERR_NO_DEVICE;
  void *d_dst = nullptr, *d_off = nullptr, *d_cf = nullptr;
  const size_t cbytes = (size_t)n * nT * nT * 2;
  int rc = M355_OK;
  if (hipMalloc(&d_dst, dst_bytes) != hipSuccess || hipMalloc(&d_off, 8 * (size_t)n) != hipSuccess || hipMalloc(&d_cf, cbytes) != hipSuccess) rc = M355_ERR_NOMEM;
  if (!rc && (hipMemcpy(d_dst, dst_base, dst_bytes, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(d_off, dst_off, 8 * (size_t)n, hipMemcpyHostToDevice) != hipSuccess ||
              hipMemcpy(d_cf, coeffs, cbytes, hipMemcpyHostToDevice) != hipSuccess)) rc = M355_ERR_HIP;
  if (!rc) {
    if (bpp == 1) hipLaunchKernelGGL(HIP_KERNEL_NAME(ks_transform_batch<uint8_t>), dim3(n), dim3(256), 0, 0, (uint8_t*)d_dst, (const long long*)d_off, stride, (const int16_t*)d_cf, log2_nT, kind, bit_depth);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(ks_transform_batch<uint16_t>), dim3(n), dim3(256), 0, 0, (uint16_t*)d_dst, (const long long*)d_off, stride, (const int16_t*)d_cf, log2_nT, kind, bit_depth);
    if (hipGetLastError() != hipSuccess || hipMemcpy(dst_base, d_dst, dst_bytes, hipMemcpyDeviceToHost) != hipSuccess) rc = M355_ERR_HIP;
  }
  hipFree(d_dst); hipFree(d_off); hipFree(d_cf);
  return rc;
}
