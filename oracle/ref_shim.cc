/*
 * ref_shim.cc — TEST INFRASTRUCTURE. Thin C-ABI access to the REAL reference (compiled from the
 * sources where they lie under /root/reference; see oracle/Makefile).  It exposes
 *   - the reference's slot table (scalar fallback, or fallback+SSE/AVX2/AVX-512 when simd!=0) with
 *     the argument shape of oracle/hevc_oracle.h so tests can call both sides identically;
 *   - sizeof/offsetof of the reference's struct acceleration_functions for the ABI-layout test.
 * The stream recorder lives in ref_recorder.cc.  This file contains no reference code.
 */
#include <stddef.h>
#include <stdint.h>
#include <string.h>
#include <stdlib.h>

#include "libde265/acceleration.h"
#include "libde265/fallback.h"
#include "libde265/fallback-dct.h"
#include "libde265/fallback-motion.h"
#include "libde265/fallback-deblk.h"
#include "libde265/intrapred.h"
#include "libde265/sps.h"
#include "libde265/x86/sse.h"

static acceleration_functions g_tab[2];
static bool g_tab_ready = false;

static const acceleration_functions& tab(int simd)
{
  if (!g_tab_ready) {
    init_acceleration_functions_fallback(&g_tab[0]);
    init_acceleration_functions_fallback(&g_tab[1]);
    init_acceleration_functions_sse(&g_tab[1]);
#if HAVE_AVX2
    init_acceleration_functions_avx2(&g_tab[1]);
#endif
#if HAVE_AVX512
    init_acceleration_functions_avx512(&g_tab[1]);
#endif
    g_tab_ready = true;
  }
  return g_tab[simd ? 1 : 0];
}

extern "C" {

int ref_shim_version(void) { return 2; }

size_t ref_accel_sizeof(void) { return sizeof(acceleration_functions); }
/* offsets of a few landmark slots, compared against struct m355_acceleration_functions */
size_t ref_accel_offsetof(int which)
{
  switch (which) {
  case 0: return offsetof(acceleration_functions, put_weighted_pred_avg_8);
  case 1: return offsetof(acceleration_functions, put_hevc_epel_8);
  case 2: return offsetof(acceleration_functions, put_hevc_qpel_8);
  case 3: return offsetof(acceleration_functions, put_hevc_qpel_16);
  case 4: return offsetof(acceleration_functions, transform_bypass);
  case 5: return offsetof(acceleration_functions, transform_add_8);
  case 6: return offsetof(acceleration_functions, transform_add_16);
  case 7: return offsetof(acceleration_functions, add_residual_8);
  case 8: return offsetof(acceleration_functions, dequant_coeff_block);
  case 9: return offsetof(acceleration_functions, deblock_luma_8);
  case 10: return offsetof(acceleration_functions, rdpcm_v);
  case 11: return offsetof(acceleration_functions, intra_pred_dc_8);
  case 12: return offsetof(acceleration_functions, intra_pred_angular_16);
  case 13: return offsetof(acceleration_functions, fwd_transform_4x4_dst_8);
  case 14: return offsetof(acceleration_functions, hadamard_transform_8);
  default: return (size_t)-1;
  }
}
/* raw access to the reference table so a test can run OUR table through the reference's own
   call pattern too */
const void* ref_accel_table(int simd) { return &tab(simd); }

/* ---- transforms ---- */
void ref_transform_add(int simd, int log2nT, int is_dst, void* dst, int pixel_bytes, const int16_t* coeffs,
                       ptrdiff_t stride, int bit_depth)
{
  const acceleration_functions& a = tab(simd);
  if (pixel_bytes == 1) {
    if (is_dst) a.transform_4x4_dst_add_8((uint8_t*)dst, coeffs, stride);
    else a.transform_add_8[log2nT - 2]((uint8_t*)dst, coeffs, stride);
  } else {
    if (is_dst) a.transform_4x4_dst_add_16((uint16_t*)dst, coeffs, stride, bit_depth);
    else a.transform_add_16[log2nT - 2]((uint16_t*)dst, coeffs, stride, bit_depth);
  }
}
void ref_transform_residual(int simd, int log2nT, int is_dst, int32_t* dst, const int16_t* coeffs, int bdShift,
                            int max_coeff_bits)
{
  const acceleration_functions& a = tab(simd);
  if (is_dst) a.transform_idst_4x4(dst, coeffs, bdShift, max_coeff_bits);
  else if (log2nT == 2) a.transform_idct_4x4(dst, coeffs, bdShift, max_coeff_bits);
  else if (log2nT == 3) a.transform_idct_8x8(dst, coeffs, bdShift, max_coeff_bits);
  else if (log2nT == 4) a.transform_idct_16x16(dst, coeffs, bdShift, max_coeff_bits);
  else a.transform_idct_32x32(dst, coeffs, bdShift, max_coeff_bits);
}
void ref_add_residual(int simd, void* dst, int pixel_bytes, ptrdiff_t stride, const int32_t* r, int nT, int bit_depth)
{
  const acceleration_functions& a = tab(simd);
  if (pixel_bytes == 1) a.add_residual_8((uint8_t*)dst, stride, r, nT, bit_depth);
  else a.add_residual_16((uint16_t*)dst, stride, r, nT, bit_depth);
}
void ref_dequant_coeff_block(int simd, int16_t* coeffBuf, const int16_t* coeffList, const int16_t* coeffPos, int nCoeff,
                             int32_t fact, int32_t offset, int32_t bdShift)
{
  tab(simd).dequant_coeff_block(coeffBuf, coeffList, coeffPos, nCoeff, fact, offset, bdShift);
}
/* which: 0 skip, 1 rdpcm_v, 2 rdpcm_h, 3 bypass, 4 bypass_rdpcm_v, 5 bypass_rdpcm_h */
void ref_residual_misc(int which, int32_t* r, const int16_t* coeffs, int nT, int tsShift, int bdShift)
{
  const acceleration_functions& a = tab(0);
  switch (which) {
  case 0: a.transform_skip_residual(r, coeffs, nT, tsShift, bdShift); break;
  case 1: a.rdpcm_v(r, coeffs, nT, tsShift, bdShift); break;
  case 2: a.rdpcm_h(r, coeffs, nT, tsShift, bdShift); break;
  case 3: a.transform_bypass(r, coeffs, nT); break;
  case 4: a.transform_bypass_rdpcm_v(r, coeffs, nT); break;
  case 5: a.transform_bypass_rdpcm_h(r, coeffs, nT); break;
  }
}
void ref_rotate_coefficients(int16_t* coeff, int nT) { tab(0).rotate_coefficients(coeff, nT); }

/* ---- motion compensation ---- */
void ref_put_qpel(int simd, int16_t* out, ptrdiff_t out_stride, const void* src, ptrdiff_t srcstride, int pixel_bytes,
                  int w, int h, int xFrac, int yFrac, int bit_depth)
{
  alignas(32) int16_t mcbuffer[64 * (64 + 7)];
  tab(simd).put_hevc_qpel(out, out_stride, src, srcstride, w, h, mcbuffer, xFrac, yFrac, bit_depth);
  (void)pixel_bytes;
}
void ref_put_epel(int simd, int16_t* out, ptrdiff_t out_stride, const void* src, ptrdiff_t srcstride, int pixel_bytes,
                  int w, int h, int xFrac, int yFrac, int bit_depth)
{
  alignas(32) int16_t mcbuffer[64 * (64 + 7)];
  const acceleration_functions& a = tab(simd);
  (void)pixel_bytes;
  /* same dispatch as mc_chroma (motion.cc:229-279) */
  if (xFrac == 0 && yFrac == 0) a.put_hevc_epel(out, out_stride, src, srcstride, w, h, 0, 0, nullptr, bit_depth);
  else if (xFrac && yFrac) a.put_hevc_epel_hv(out, out_stride, src, srcstride, w, h, xFrac, yFrac, mcbuffer, bit_depth);
  else if (xFrac) a.put_hevc_epel_h(out, out_stride, src, srcstride, w, h, xFrac, yFrac, mcbuffer, bit_depth);
  else a.put_hevc_epel_v(out, out_stride, src, srcstride, w, h, xFrac, yFrac, mcbuffer, bit_depth);
}
void ref_put_unweighted_pred(int simd, void* dst, ptrdiff_t ds, int pb, const int16_t* src, ptrdiff_t ss, int w, int h,
                             int bd)
{
  (void)pb;
  tab(simd).put_unweighted_pred(dst, ds, src, ss, w, h, bd);
}
void ref_put_weighted_pred_avg(int simd, void* dst, ptrdiff_t ds, int pb, const int16_t* s1, const int16_t* s2,
                               ptrdiff_t ss, int w, int h, int bd)
{
  (void)pb;
  tab(simd).put_weighted_pred_avg(dst, ds, s1, s2, ss, w, h, bd);
}
void ref_put_weighted_pred(int simd, void* dst, ptrdiff_t ds, int pb, const int16_t* src, ptrdiff_t ss, int w, int h,
                           int wt, int o, int log2WD, int bd)
{
  (void)pb;
  tab(simd).put_weighted_pred(dst, ds, src, ss, w, h, wt, o, log2WD, bd);
}
void ref_put_weighted_bipred(int simd, void* dst, ptrdiff_t ds, int pb, const int16_t* s1, const int16_t* s2,
                             ptrdiff_t ss, int w, int h, int w1, int o1, int w2, int o2, int log2WD, int bd)
{
  (void)pb;
  tab(simd).put_weighted_bipred(dst, ds, s1, s2, ss, w, h, w1, o1, w2, o2, log2WD, bd);
}

/* ---- intra ---- */
void ref_intra_pred(int simd, int which /*0 planar,1 dc,2 angular*/, void* dst, ptrdiff_t stride, int pb, int bit_depth,
                    int disableBoundaryFilter, int mode, int nT, int cIdx, const void* border)
{
  const acceleration_functions& a = tab(simd);
  if (pb == 1) {
    if (which == 0) a.intra_pred_planar_8((uint8_t*)dst, stride, nT, cIdx, (const uint8_t*)border);
    else if (which == 1) a.intra_pred_dc_8((uint8_t*)dst, stride, nT, cIdx, (const uint8_t*)border);
    else a.intra_pred_angular_8((uint8_t*)dst, stride, bit_depth, disableBoundaryFilter, 0, 0, mode, nT, cIdx,
                                (const uint8_t*)border);
  } else {
    if (which == 0) a.intra_pred_planar_16((uint16_t*)dst, stride, nT, cIdx, (const uint16_t*)border);
    else if (which == 1) a.intra_pred_dc_16((uint16_t*)dst, stride, nT, cIdx, (const uint16_t*)border);
    else a.intra_pred_angular_16((uint16_t*)dst, stride, bit_depth, disableBoundaryFilter, 0, 0, mode, nT, cIdx,
                                 (const uint16_t*)border);
  }
}
void ref_intra_sample_filtering(void* border, int pb, int nT, int cIdx, int mode, int strong, int bd_luma)
{
  static seq_parameter_set* sps = nullptr;
  if (!sps) sps = new seq_parameter_set;
  sps->strong_intra_smoothing_enable_flag = strong;
  sps->bit_depth_luma = bd_luma;
  if (pb == 1) intra_prediction_sample_filtering<uint8_t>(*sps, (uint8_t*)border, nT, cIdx, (enum IntraPredMode)mode);
  else intra_prediction_sample_filtering<uint16_t>(*sps, (uint16_t*)border, nT, cIdx, (enum IntraPredMode)mode);
}

/* ---- deblocking kernels ---- */
void ref_deblock_luma(int simd, void* ptr, ptrdiff_t stride, int pb, int vertical, int dE, int dEp, int dEq, int tc,
                      int filterP, int filterQ, int bd)
{
  if (pb == 1) tab(simd).deblock_luma_8((uint8_t*)ptr, stride, vertical, dE, dEp, dEq, tc, filterP, filterQ);
  else deblock_luma_kernel<uint16_t>((uint16_t*)ptr, stride, vertical != 0, dE, dEp, dEq, tc, filterP != 0, filterQ != 0, bd);
}
void ref_deblock_chroma(int simd, void* ptr, ptrdiff_t stride, int pb, int vertical, int tc, int filterP, int filterQ,
                        int bd)
{
  if (pb == 1) tab(simd).deblock_chroma_8((uint8_t*)ptr, stride, vertical, tc, filterP, filterQ);
  else deblock_chroma_kernel<uint16_t>((uint16_t*)ptr, stride, vertical != 0, tc, filterP != 0, filterQ != 0, bd);
}

} // extern "C"

/* Layer a backend's table initialiser on a live decoder's acceleration table, exactly where the reference layers its own
 * SSE / AVX / ARM tables (base_context::set_acceleration_functions, decctx.cc:239-270: fallback first, then the overrides).
 * `init` has the signature of init_acceleration_functions_mi355x (include/de265_mi355x.h).  Test infrastructure. */
#include "libde265/decctx.h"
extern "C" __attribute__((visibility("default")))
int ref_layer_acceleration(de265_decoder_context* c, int (*init)(void*))
{
  decoder_context* ctx = (decoder_context*)c;
  ctx->set_acceleration_functions(de265_acceleration_SCALAR);
  return init(&ctx->acceleration);
}
