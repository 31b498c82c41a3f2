/*
 * hevc_oracle.h — CPU restatement of libde265's pixel-reconstruction path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is linked, imported or executed by the product
 * library (libde265_amd/): only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use
 * it, and only as the checker.  Parity of this restatement is PINNED against the reference itself:
 *   - slot level: every o_* kernel vs the reference's fallback-* functions compiled from
 *     /root/reference (oracle/_ref/libde265_ref.so via ref_shim.cc), tests/test_oracle_vs_ref.py;
 *   - picture level: work lists recorded from the reference decoder on testdata/girlshy.h265
 *     replayed through o_decode_picture() must reproduce the reference's frames and the CI golden
 *     MD5 b81538fa33a67278e5263e231e43ca98 (scripts/ci-run.sh:91-92), tests/test_girlshy.py.
 *
 * Each function cites the reference file:line it follows.
 */
#ifndef HEVC_ORACLE_H
#define HEVC_ORACLE_H

#include <stddef.h>
#include <stdint.h>
#include "de265_mi355x.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- slot-level restatements (same signatures as libde265/acceleration.h slots) ---- */

/* fallback-dct.cc:550-691 / :269-407 */
void o_transform_add_8(int log2nT, uint8_t* dst, const int16_t* coeffs, ptrdiff_t stride);
void o_transform_add_16(int log2nT, uint16_t* dst, const int16_t* coeffs, ptrdiff_t stride, int bit_depth);
void o_transform_4x4_dst_add_8(uint8_t* dst, const int16_t* coeffs, ptrdiff_t stride);
void o_transform_4x4_dst_add_16(uint16_t* dst, const int16_t* coeffs, ptrdiff_t stride, int bit_depth);
/* fallback-dct.cc:469-508, 695-859 */
void o_transform_idst_4x4(int32_t* dst, const int16_t* coeffs, int bdShift, int max_coeff_bits);
void o_transform_idct(int log2nT, int32_t* dst, const int16_t* coeffs, int bdShift, int max_coeff_bits);
/* fallback-dct.h:65-73 */
void o_add_residual_8(uint8_t* dst, ptrdiff_t stride, const int32_t* r, int nT, int bit_depth);
void o_add_residual_16(uint16_t* dst, ptrdiff_t stride, const int32_t* r, int nT, int bit_depth);
/* fallback-dct.cc:1212-1220 */
void o_dequant_coeff_block(int16_t* coeffBuf, const int16_t* coeffList, const int16_t* coeffPos,
                           int nCoeff, int32_t fact, int32_t offset, int32_t bdShift);
/* fallback-dct.cc:81-91, 187-225, 161-185, 228-256 */
void o_transform_skip_residual(int32_t* residual, const int16_t* coeffs, int nT, int tsShift, int bdShift);
void o_rdpcm_v(int32_t* residual, const int16_t* coeffs, int nT, int tsShift, int bdShift);
void o_rdpcm_h(int32_t* residual, const int16_t* coeffs, int nT, int tsShift, int bdShift);
void o_transform_bypass(int32_t* r, const int16_t* coeffs, int nT);
void o_transform_bypass_rdpcm_v(int32_t* r, const int16_t* coeffs, int nT);
void o_transform_bypass_rdpcm_h(int32_t* r, const int16_t* coeffs, int nT);
void o_rotate_coefficients(int16_t* coeff, int nT);

/* fallback-motion.cc:492-636, 431-485 (qpel) and :305-415, 262-302 (epel).  pixel_bytes 1|2. */
void o_put_qpel(int16_t* out, ptrdiff_t out_stride, const void* src, ptrdiff_t srcstride, int pixel_bytes,
                int nPbW, int nPbH, int xFracL, int yFracL, int bit_depth);
void o_put_epel(int16_t* out, ptrdiff_t out_stride, const void* src, ptrdiff_t srcstride, int pixel_bytes,
                int nPbWC, int nPbHC, int xFracC, int yFracC, int bit_depth);
/* fallback-motion.cc:33-256 */
void o_put_unweighted_pred(void* dst, ptrdiff_t dststride, int pixel_bytes, const int16_t* src,
                           ptrdiff_t srcstride, int w, int h, int bit_depth);
void o_put_weighted_pred_avg(void* dst, ptrdiff_t dststride, int pixel_bytes, const int16_t* src1,
                             const int16_t* src2, ptrdiff_t srcstride, int w, int h, int bit_depth);
void o_put_weighted_pred(void* dst, ptrdiff_t dststride, int pixel_bytes, const int16_t* src,
                         ptrdiff_t srcstride, int w, int h, int wt, int o, int log2WD, int bit_depth);
void o_put_weighted_bipred(void* dst, ptrdiff_t dststride, int pixel_bytes, const int16_t* src1,
                           const int16_t* src2, ptrdiff_t srcstride, int w, int h, int w1, int o1,
                           int w2, int o2, int log2WD, int bit_depth);

/* intrapred.h:261-433 ; border points at element 0 of p[-2nT..2nT] */
void o_intra_pred_dc(void* dst, ptrdiff_t stride, int pixel_bytes, int nT, int cIdx, const void* border);
void o_intra_pred_planar(void* dst, ptrdiff_t stride, int pixel_bytes, int nT, int cIdx, const void* border);
void o_intra_pred_angular(void* dst, ptrdiff_t stride, int pixel_bytes, int bit_depth,
                          int disableBoundaryFilter, int mode, int nT, int cIdx, const void* border);
/* intrapred.h:185-258 (in place on border) */
void o_intra_sample_filtering(void* border, int pixel_bytes, int nT, int cIdx, int mode,
                              int strong_intra_smoothing, int bit_depth_luma);

/* fallback-deblk.h:33-124 */
void o_deblock_luma(void* ptr, ptrdiff_t stride, int pixel_bytes, int vertical, int dE, int dEp, int dEq,
                    int tc, int filterP, int filterQ, int bit_depth);
void o_deblock_chroma(void* ptr, ptrdiff_t stride, int pixel_bytes, int vertical, int tc, int filterP,
                      int filterQ, int bit_depth);

/* ---- picture level ---- */

typedef struct o_frame {
  int width, height, chroma_format, bd_luma, bd_chroma;
  int w[3], h[3];
  ptrdiff_t stride[3];
  uint16_t* p[3];          /* samples are held as uint16 whatever the bit depth */
} o_frame;

o_frame* o_frame_new(int width, int height, int chroma_format, int bd_luma, int bd_chroma);
void o_frame_free(o_frame* f);
/* stride in samples; bytes_per_sample 1 or 2 */
void o_frame_import(o_frame* f, int cidx, const void* src, ptrdiff_t stride, int bytes_per_sample);
void o_frame_export(const o_frame* f, int cidx, void* dst, ptrdiff_t stride, int bytes_per_sample);

/* Decode one picture's work lists into dst (pic->dst_frame / ref_frames handles are ignored; refs[]
 * is indexed by m355_pb.ref_slot). stages = M355_STAGE_* mask. Returns 0 or M355_ERR_INVALID. */
int o_decode_picture(const m355_picture* pic, o_frame* dst, o_frame* const* refs, int stages);

/* ---- SEI decoded picture hash of one plane (sei.cc:161-257); data = w x h samples (uint8_t for bit_depth <= 8,
 * else uint16_t), stride in samples ---- */
uint32_t o_hash_checksum(const void* data, int w, int h, ptrdiff_t stride, int bit_depth);
uint32_t o_hash_crc(const void* data, int w, int h, ptrdiff_t stride, int bit_depth);
void o_hash_md5(const void* data, int w, int h, ptrdiff_t stride, int bit_depth, uint8_t out[16]);

#ifdef __cplusplus
}
#endif
#endif
