/*
 * de265_mi355x.h — C ABI of the MI355X (gfx950 / CDNA4) HEVC pixel-reconstruction backend.
 *
 * This is the drop-in boundary for the ONE hot path of strukturag/libde265 that this project
 * accelerates: everything behind `struct acceleration_functions` (libde265/acceleration.h:29-231)
 * plus the two in-loop filter drivers that bypass that table (libde265/deblock.cc:908-946,
 * libde265/sao.cc:327-382).  Two layers are exported, both `extern "C"`, plain pointers/sizes only:
 *
 *  (1) SLOT LAYER  — `init_acceleration_functions_mi355x()` fills a table with the EXACT layout and
 *      signatures of the reference's `struct acceleration_functions` (cf. the reference's own
 *      `init_acceleration_functions_fallback` libde265/fallback.cc:28 and `_sse` x86/sse.cc:46).
 *      Every slot is synchronous, takes host pointers, and runs the HIP kernel for one block.
 *      It exists for parity (the tests drive it exactly like dev-tools/test-*.cc drive the SIMD
 *      tables) and as the trivially-correct entry; it is NOT the fast path (one PCIe round trip
 *      per block).  `m355_*_batch()` variants run N blocks per launch.
 *
 *  (2) PICTURE LAYER — the deferred reconstruction executor.  The host parser (the reference's own
 *      slice.cc / motion.cc glue) RECORDS per-picture work lists instead of
 *      calling slots per block; `m355_submit_picture()` uploads them and runs
 *         inter prediction → residual (dequant + IDCT/IDST/skip/bypass) → intra wavefront →
 *         deblock V → deblock H → SAO
 *      on the device, against reference frames that stay resident in HBM.
 *
 * All structs are plain little-endian PODs shared by the HIP library, the reference-side glue that
 * records them inside the reference's decoder (glue/m355_glue.cc), the CPU oracle
 * (oracle/hevc_oracle.c) and the Python test layer (numpy dtypes in libde265_amd/worklist.py mirror
 * them field by field; tests/test_abi_layout.py checks sizes).
 */
#ifndef DE265_MI355X_H
#define DE265_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define M355_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------------------------------
 * Error codes (the reference's slots return void and have no error channel, acceleration.h:29-231;
 * the picture layer can fail on resources, so it reports).
 * ---------------------------------------------------------------------------------------------- */
enum {
  M355_OK = 0,
  M355_ERR_NO_DEVICE = 1,     /* no HIP device / runtime failure at init            */
  M355_ERR_HIP = 2,           /* a HIP call failed; m355_last_error() has the text   */
  M355_ERR_INVALID = 3,       /* malformed work list / bad argument                  */
  M355_ERR_NOMEM = 4,
  M355_ERR_TIMEOUT = 5,       /* intra wavefront spin bound exceeded (device flag)   */
  M355_ERR_BUSY = 6,          /* m355_decode_status: the decode has not finished yet */
  M355_ERR_STALE = 7          /* m355_decode_status: the serial is older than the status ring (its outcome is no longer kept) */
};

M355_API const char* m355_last_error(void);
M355_API int m355_device_count(void);
M355_API const char* m355_version(void);

/* ================================================================================================
 * (1) SLOT LAYER — same member order as libde265/acceleration.h:29-231 (94 pointers, 752 bytes on
 * x86-64; the reference struct's inline wrapper methods/templates occupy no storage).
 * ============================================================================================== */

struct m355_acceleration_functions {
  /* weighted prediction write-back, 8 bit (acceleration.h:31-47) */
  void (*put_weighted_pred_avg_8)(uint8_t* dst, ptrdiff_t dststride, const int16_t* src1,
                                  const int16_t* src2, ptrdiff_t srcstride, int width, int height);
  void (*put_unweighted_pred_8)(uint8_t* dst, ptrdiff_t dststride, const int16_t* src,
                                ptrdiff_t srcstride, int width, int height);
  void (*put_weighted_pred_8)(uint8_t* dst, ptrdiff_t dststride, const int16_t* src,
                              ptrdiff_t srcstride, int width, int height, int w, int o, int log2WD);
  void (*put_weighted_bipred_8)(uint8_t* dst, ptrdiff_t dststride, const int16_t* src1,
                                const int16_t* src2, ptrdiff_t srcstride, int width, int height,
                                int w1, int o1, int w2, int o2, int log2WD);
  /* 9-16 bit (acceleration.h:50-64) */
  void (*put_weighted_pred_avg_16)(uint16_t* dst, ptrdiff_t dststride, const int16_t* src1,
                                   const int16_t* src2, ptrdiff_t srcstride, int width, int height,
                                   int bit_depth);
  void (*put_unweighted_pred_16)(uint16_t* dst, ptrdiff_t dststride, const int16_t* src,
                                 ptrdiff_t srcstride, int width, int height, int bit_depth);
  void (*put_weighted_pred_16)(uint16_t* dst, ptrdiff_t dststride, const int16_t* src,
                               ptrdiff_t srcstride, int width, int height, int w, int o, int log2WD,
                               int bit_depth);
  void (*put_weighted_bipred_16)(uint16_t* dst, ptrdiff_t dststride, const int16_t* src1,
                                 const int16_t* src2, ptrdiff_t srcstride, int width, int height,
                                 int w1, int o1, int w2, int o2, int log2WD, int bit_depth);

  /* chroma (epel) and luma (qpel) interpolation, 8 bit (acceleration.h:87-102).
     NOTE: as in the reference, put_hevc_epel_8 has NO bit_depth argument, the others do. */
  void (*put_hevc_epel_8)(int16_t* dst, ptrdiff_t dststride, const uint8_t* src, ptrdiff_t srcstride,
                          int width, int height, int mx, int my, int16_t* mcbuffer);
  void (*put_hevc_epel_h_8)(int16_t* dst, ptrdiff_t dststride, const uint8_t* src,
                            ptrdiff_t srcstride, int width, int height, int mx, int my,
                            int16_t* mcbuffer, int bit_depth);
  void (*put_hevc_epel_v_8)(int16_t* dst, ptrdiff_t dststride, const uint8_t* src,
                            ptrdiff_t srcstride, int width, int height, int mx, int my,
                            int16_t* mcbuffer, int bit_depth);
  void (*put_hevc_epel_hv_8)(int16_t* dst, ptrdiff_t dststride, const uint8_t* src,
                             ptrdiff_t srcstride, int width, int height, int mx, int my,
                             int16_t* mcbuffer, int bit_depth);
  void (*put_hevc_qpel_8[4][4])(int16_t* dst, ptrdiff_t dststride, const uint8_t* src,
                                ptrdiff_t srcstride, int width, int height, int16_t* mcbuffer);
  /* 9-16 bit (acceleration.h:105-120) */
  void (*put_hevc_epel_16)(int16_t* dst, ptrdiff_t dststride, const uint16_t* src,
                           ptrdiff_t srcstride, int width, int height, int mx, int my,
                           int16_t* mcbuffer, int bit_depth);
  void (*put_hevc_epel_h_16)(int16_t* dst, ptrdiff_t dststride, const uint16_t* src,
                             ptrdiff_t srcstride, int width, int height, int mx, int my,
                             int16_t* mcbuffer, int bit_depth);
  void (*put_hevc_epel_v_16)(int16_t* dst, ptrdiff_t dststride, const uint16_t* src,
                             ptrdiff_t srcstride, int width, int height, int mx, int my,
                             int16_t* mcbuffer, int bit_depth);
  void (*put_hevc_epel_hv_16)(int16_t* dst, ptrdiff_t dststride, const uint16_t* src,
                              ptrdiff_t srcstride, int width, int height, int mx, int my,
                              int16_t* mcbuffer, int bit_depth);
  void (*put_hevc_qpel_16[4][4])(int16_t* dst, ptrdiff_t dststride, const uint16_t* src,
                                 ptrdiff_t srcstride, int width, int height, int16_t* mcbuffer,
                                 int bit_depth);

  /* inverse transforms (acceleration.h:143-170) */
  void (*transform_bypass)(int32_t* residual, const int16_t* coeffs, int nT);
  void (*transform_bypass_rdpcm_v)(int32_t* r, const int16_t* coeffs, int nT);
  void (*transform_bypass_rdpcm_h)(int32_t* r, const int16_t* coeffs, int nT);
  void (*transform_skip_8)(uint8_t* dst, const int16_t* coeffs, ptrdiff_t stride);          /* dead slot in the reference (fallback-dct.cc:52) */
  void (*transform_skip_rdpcm_v_8)(uint8_t* dst, const int16_t* coeffs, int nT, ptrdiff_t stride);
  void (*transform_skip_rdpcm_h_8)(uint8_t* dst, const int16_t* coeffs, int nT, ptrdiff_t stride);
  void (*transform_4x4_dst_add_8)(uint8_t* dst, const int16_t* coeffs, ptrdiff_t stride);
  void (*transform_add_8[4])(uint8_t* dst, const int16_t* coeffs, ptrdiff_t stride);
  void (*transform_skip_16)(uint16_t* dst, const int16_t* coeffs, ptrdiff_t stride, int bit_depth); /* dead slot (fallback-dct.cc:69) */
  void (*transform_4x4_dst_add_16)(uint16_t* dst, const int16_t* coeffs, ptrdiff_t stride,
                                   int bit_depth);
  void (*transform_add_16[4])(uint16_t* dst, const int16_t* coeffs, ptrdiff_t stride, int bit_depth);
  void (*rotate_coefficients)(int16_t* coeff, int nT);
  void (*transform_idst_4x4)(int32_t* dst, const int16_t* coeffs, int bdShift, int max_coeff_bits);
  void (*transform_idct_4x4)(int32_t* dst, const int16_t* coeffs, int bdShift, int max_coeff_bits);
  void (*transform_idct_8x8)(int32_t* dst, const int16_t* coeffs, int bdShift, int max_coeff_bits);
  void (*transform_idct_16x16)(int32_t* dst, const int16_t* coeffs, int bdShift, int max_coeff_bits);
  void (*transform_idct_32x32)(int32_t* dst, const int16_t* coeffs, int bdShift, int max_coeff_bits);
  void (*add_residual_8)(uint8_t* dst, ptrdiff_t stride, const int32_t* r, int nT, int bit_depth);
  void (*add_residual_16)(uint16_t* dst, ptrdiff_t stride, const int32_t* r, int nT, int bit_depth);

  /* dequantisation (acceleration.h:175-181) */
  void (*dequant_coeff_block)(int16_t* coeffBuf, const int16_t* coeffList, const int16_t* coeffPos,
                              int nCoeff, int32_t fact, int32_t offset, int32_t bdShift);

  /* deblocking, one 4-line edge segment (acceleration.h:184-187) */
  void (*deblock_luma_8)(uint8_t* ptr, ptrdiff_t stride, int vertical, int dE, int dEp, int dEq,
                         int tc, int filterP, int filterQ);
  void (*deblock_chroma_8)(uint8_t* ptr, ptrdiff_t stride, int vertical, int tc, int filterP,
                           int filterQ);

  /* (R)DPCM / transform skip residuals (acceleration.h:189-193) */
  void (*rdpcm_v)(int32_t* residual, const int16_t* coeffs, int nT, int tsShift, int bdShift);
  void (*rdpcm_h)(int32_t* residual, const int16_t* coeffs, int nT, int tsShift, int bdShift);
  void (*transform_skip_residual)(int32_t* residual, const int16_t* coeffs, int nT, int tsShift,
                                  int bdShift);

  /* intra prediction from a prepared border[-2nT..2nT] (acceleration.h:205-212) */
  void (*intra_pred_dc_8)(uint8_t* dst, ptrdiff_t stride, int nT, int cIdx, const uint8_t* border);
  void (*intra_pred_dc_16)(uint16_t* dst, ptrdiff_t stride, int nT, int cIdx, const uint16_t* border);
  void (*intra_pred_planar_8)(uint8_t* dst, ptrdiff_t stride, int nT, int cIdx,
                              const uint8_t* border);
  void (*intra_pred_planar_16)(uint16_t* dst, ptrdiff_t stride, int nT, int cIdx,
                               const uint16_t* border);
  void (*intra_pred_angular_8)(uint8_t* dst, ptrdiff_t stride, int bit_depth,
                               int disableBoundaryFilter, int xB0, int yB0, int mode, int nT,
                               int cIdx, const uint8_t* border);
  void (*intra_pred_angular_16)(uint16_t* dst, ptrdiff_t stride, int bit_depth,
                                int disableBoundaryFilter, int xB0, int yB0, int mode, int nT,
                                int cIdx, const uint16_t* border);

  /* forward transforms — encoder only (acceleration.h:222-230); OUT OF SCOPE: left untouched
     (the fallback's pointers stay in place when the table was pre-filled, NULL otherwise). */
  void (*fwd_transform_4x4_dst_8)(int16_t* coeffs, const int16_t* src, ptrdiff_t stride);
  void (*fwd_transform_8[4])(int16_t* coeffs, const int16_t* src, ptrdiff_t stride);
  void (*hadamard_transform_8[4])(int16_t* coeffs, const int16_t* src, ptrdiff_t stride);
};

/* Replaces: init_acceleration_functions_sse/_avx2/_avx512 (libde265/x86/sse.cc:46-171) at the call
 * site libde265/decctx.cc:239-270 — call it AFTER init_acceleration_functions_fallback(); it
 * overrides every decoder slot and leaves the encoder-only forward transforms untouched.
 * `accel` must point to a `struct acceleration_functions` (identical layout). Returns M355_OK, or
 * M355_ERR_NO_DEVICE without touching the table (fail loudly; there is no CPU fallback here). */
M355_API int init_acceleration_functions_mi355x(void* accel);

/* ------------------------------------------------------------------------------------------------
 * Batched slot entry points: N independent blocks per launch, host pointers.
 * ---------------------------------------------------------------------------------------------- */

/* kind: 0 = IDCT (transform_add[log2-2]), 1 = 4x4 IDST (transform_4x4_dst_add). dst blocks live in
 * one host buffer `dst_base` (byte size dst_bytes) at sample offsets dst_off[i], row stride in
 * samples; coeffs are dense nT*nT int16 per block, consecutive. pixel_bytes 1 (8 bit) or 2. */
M355_API int m355_transform_add_batch(int n, int log2_nT, int kind, int bit_depth,
                                      void* dst_base, size_t dst_bytes, const int64_t* dst_off,
                                      ptrdiff_t stride, const int16_t* coeffs);

/* ================================================================================================
 * (2) PICTURE LAYER — work lists.
 * ============================================================================================== */

#define M355_MAX_TILE_COLS 20   /* DE265_MAX_TILE_COLUMNS, libde265/pps.h:30 */
#define M355_MAX_TILE_ROWS 22   /* DE265_MAX_TILE_ROWS,    libde265/pps.h:31 */
#define M355_MAX_REF_FRAMES 32  /* DPB slots addressable by one picture      */

/* picture-level flags (pic_params.flags) */
enum {
  M355_PF_CONSTRAINED_INTRA_PRED   = 1 << 0,  /* pps.constrained_intra_pred_flag (intrapred.h:546)      */
  M355_PF_STRONG_INTRA_SMOOTHING   = 1 << 1,  /* sps.strong_intra_smoothing_enable_flag (intrapred.h:216)*/
  M355_PF_PCM_LOOP_FILTER_DISABLE  = 1 << 2,  /* sps.pcm_loop_filter_disable_flag (deblock.cc:578)       */
  M355_PF_LF_ACROSS_TILES          = 1 << 3,  /* pps.loop_filter_across_tiles_enabled_flag               */
  M355_PF_SAO_ENABLED              = 1 << 4,  /* sps.sample_adaptive_offset_enabled_flag && !DISABLE_SAO */
  M355_PF_INTRA_SMOOTHING_DISABLED = 1 << 5,  /* sps.range_extension.intra_smoothing_disabled_flag       */
  M355_PF_IMPLICIT_RDPCM           = 1 << 6,  /* sps.range_extension.implicit_rdpcm_enabled_flag         */
  M355_PF_SCALING_LIST             = 1 << 7,  /* sps.scaling_list_enable_flag (transform.cc:461)         */
  M355_PF_DEBLOCK_ENABLED          = 1 << 8,  /* !DE265_DECODER_PARAM_DISABLE_DEBLOCKING (de265.h:409)   */
  M355_PF_CROSS_COMPONENT_PRED     = 1 << 9,  /* pps.range_extension.cross_component_prediction_enabled_flag
                                                 (4:4:4 only, transform.cc:244-260): chroma blocks may carry a
                                                 ResScaleVal in m355_rb.matrix_id (see there)            */
  M355_PF_TRANSFORM_SKIP_ROTATION  = 1 << 10, /* sps.range_extension.transform_skip_rotation_enabled_flag (transform.cc:400-402);
                                                 informative: the kernels follow the per-block M355_RBF_ROTATE */
  M355_PF_CLEAR_DST                = 1 << 11  /* zero the picture before reconstructing it, as the reference's allocator does for
                                                 every new picture (image.cc:164): only matters when the lists do not cover the
                                                 whole picture (damaged streams: missing slices, PBs without a usable list) */
};

typedef struct m355_pic_params {
  int32_t  width, height;          /* pic_width/height_in_luma_samples                      */
  uint8_t  chroma_format_idc;      /* 0 mono, 1 4:2:0, 2 4:2:2, 3 4:4:4                     */
  uint8_t  bit_depth_luma, bit_depth_chroma;
  uint8_t  log2_ctb_size;          /* Log2CtbSizeY                                          */
  uint8_t  log2_min_tb_size;       /* Log2MinTrafoSize (z-scan granularity, pps.cc:608-623) */
  uint8_t  log2_min_cb_size;       /* Log2MinCbSizeY                                        */
  int8_t   pic_cb_qp_offset, pic_cr_qp_offset;   /* deblock.cc:666-668                      */
  uint32_t flags;                  /* M355_PF_*                                             */
  uint8_t  num_tile_cols, num_tile_rows;         /* 1,1 when tiles are off                  */
  uint16_t col_bd[M355_MAX_TILE_COLS + 1];       /* pps.colBd, in CTBs                      */
  uint16_t row_bd[M355_MAX_TILE_ROWS + 1];       /* pps.rowBd                               */
  uint16_t reserved;
} m355_pic_params;                 /* 112 bytes */

/* slice (segment) header fields the pixel path reads */
enum {
  M355_SF_DEBLOCK_DISABLED = 1 << 0,  /* slice_deblocking_filter_disabled_flag (deblock.cc:216)        */
  M355_SF_LF_ACROSS_SLICES = 1 << 1,  /* slice_loop_filter_across_slices_enabled_flag                  */
  M355_SF_SAO_LUMA         = 1 << 2,  /* slice_sao_luma_flag (sao.cc:367)                              */
  M355_SF_SAO_CHROMA       = 1 << 3   /* slice_sao_chroma_flag                                         */
};
typedef struct m355_slice {
  int32_t slice_addr_rs;           /* SliceAddrRS */
  int8_t  beta_offset, tc_offset;  /* slice_beta_offset, slice_tc_offset (deblock.cc:521-523) */
  uint8_t flags;                   /* M355_SF_* */
  uint8_t reserved;
} m355_slice;                      /* 8 bytes */

enum { M355_CTBF_HAS_PCM_OR_BYPASS = 1 << 0 };  /* image.h get_CTB_has_pcm_or_cu_transquant_bypass */
typedef struct m355_ctb {
  uint16_t slice_idx;              /* index into slices[] (ctb_info.SliceHeaderIndex)              */
  uint8_t  sao_type;               /* sao_info.SaoTypeIdx  — 2 bits per cIdx (slice.h:271)         */
  uint8_t  sao_eo_class;           /* sao_info.SaoEoClass  — 2 bits per cIdx                       */
  uint8_t  sao_band_pos[3];        /* sao_info.sao_band_position                                   */
  uint8_t  flags;                  /* M355_CTBF_*                                                  */
  int8_t   sao_offset[3][4];       /* sao_info.saoOffsetVal                                        */
  uint32_t ib_start, ib_count;     /* this CTB's range in ibs[] (decode order)                     */
} m355_ctb;                        /* 28 bytes */

/* coding unit: rasterised on the device into the per-min-CB plane (image.h:173-195 CB_ref_info) */
enum { M355_CUF_PCM = 1 << 0, M355_CUF_TRANSQUANT_BYPASS = 1 << 1 };
typedef struct m355_cu {
  uint16_t x, y;                   /* luma position */
  uint8_t  log2_size;
  uint8_t  pred_mode;              /* 0 MODE_INTRA, 1 MODE_INTER, 2 MODE_SKIP (slice.h:87-90) */
  uint8_t  part_mode;              /* PartMode (slice.h:72-82) */
  int8_t   qp_y;                   /* QP_Y (transform.cc:199) */
  uint8_t  flags;                  /* M355_CUF_* */
  uint8_t  reserved[3];
} m355_cu;                         /* 12 bytes */

/* transform-tree leaf (luma geometry): transform edges + cbf_luma for deblocking
 * (deblock.cc:33-63 markTransformBlockBoundary, slice.cc:2958-2960 set_nonzero_coefficient) */
enum { M355_TUF_NONZERO_COEFF = 1 << 0 };
typedef struct m355_tu {
  uint16_t x, y;
  uint8_t  log2_size;
  uint8_t  flags;
  uint16_t reserved;
} m355_tu;                         /* 8 bytes */

/* explicit weighted-prediction entry: one (list, refIdx) of one slice (motion.cc:520-529) */
typedef struct m355_wt {
  int16_t w[3];                    /* LumaWeight / ChromaWeight[..][0] / [..][1]                      */
  int16_t o[3];                    /* offsets already << WpOffsetBdShift (motion.cc:524)              */
  uint8_t log2wd_luma;             /* luma_log2_weight_denom + shift1_L (motion.cc:520)               */
  uint8_t log2wd_chroma;
  uint16_t reserved;
} m355_wt;                         /* 16 bytes */

/* prediction block (motion.cc:288 generate_inter_prediction_samples, one call) */
enum {
  M355_PBF_PRED_L0   = 1 << 0,     /* PBMotion.predFlag[0] as stored (used by deblock bS)         */
  M355_PBF_PRED_L1   = 1 << 1,
  M355_PBF_MC_L0     = 1 << 2,     /* list actually interpolated (after motion.cc:348-357 demotion) */
  M355_PBF_MC_L1     = 1 << 3,
  M355_PBF_WEIGHTED  = 1 << 4,     /* explicit weights (motion.cc:508,571,633) */
  M355_PBF_FILL_L0   = 1 << 5,     /* reference missing → predSamples = 1<<13 (motion.cc:362-376) */
  M355_PBF_FILL_L1   = 1 << 6
};
typedef struct m355_pb {
  uint16_t x, y;                   /* luma position xP,yP */
  uint8_t  w, h;                   /* nPbW, nPbH (luma) */
  uint8_t  flags;                  /* M355_PBF_* */
  uint8_t  reserved;
  int8_t   ref_slot[2];            /* index into m355_picture.ref_frames (DPB identity), -1 none */
  int16_t  mv[2][2];               /* [list][x,y] quarter-pel */
  uint16_t wt_idx[2];              /* index into wts[] per list (valid when WEIGHTED) */
  uint16_t reserved2;
} m355_pb;                         /* 24 bytes */

/* residual block: one scale_coefficients() call (transform.cc:645 / :361-642) */
enum {
  M355_RK_DCT = 0, M355_RK_DST = 1, M355_RK_SKIP = 2, M355_RK_BYPASS = 3
};
enum {
  M355_RBF_DEFERRED  = 1 << 0,     /* intra block: store residual to the residual buffer (res_ofs)
                                      instead of adding it to the picture                         */
  M355_RBF_RDPCM_H   = 1 << 1,     /* rdpcmMode 1 */
  M355_RBF_RDPCM_V   = 1 << 2,     /* rdpcmMode 2 */
  M355_RBF_ROTATE    = 1 << 3,     /* transform_skip_rotation (transform.cc:400-402) */
  M355_RBF_DEQUANTIZED = 1 << 4    /* coeffs[] levels are already scaled (what the table slots
                                      transform_add / transform_skip_residual receive): skip dequant */
};
typedef struct m355_rb {
  uint16_t x, y;                   /* position in component samples */
  uint8_t  cidx;
  uint8_t  log2_size;              /* 2..5 */
  uint8_t  kind;                   /* M355_RK_* */
  uint8_t  flags;                  /* M355_RBF_* */
  uint8_t  qp;                     /* qPYPrime / qPCbPrime / qPCrPrime (transform.cc:371-377) */
  uint8_t  matrix_id;              /* bits 0-2: scaling-list matrixID (transform.cc:493-502), unused if no list.
                                      Cross-component prediction (M355_PF_CROSS_COMPONENT_PRED, chroma blocks): bits 4-6 =
                                      log2_res_scale_abs_plus1 (0 = none), bit 7 = sign -> ResScaleVal = +-(1 << (v-1))
                                      (slice.cc:3721-3760); bit 3 = the TU's luma block is rbs[i-2] (else rbs[i-1]): the
                                      three blocks of a 4:4:4 transform unit have one size, so they sit next to each other
                                      in their size bin.  A chroma block with cbf 0 but ResScaleVal != 0 is listed with
                                      ncoeff 0 (slice.cc:3512-3523). */
  uint16_t ncoeff;
  uint32_t coeff_ofs;              /* first entry in coeffs[] */
  uint32_t res_ofs;                /* int16 offset in the residual buffer (DEFERRED) */
} m355_rb;                         /* 20 bytes */

/* intra block: one decode_intra_prediction() call (intrapred.cc:321), in decode order */
enum {
  M355_IBF_HAS_RESIDUAL            = 1 << 0,
  M355_IBF_DISABLE_BOUNDARY_FILTER = 1 << 1,  /* intrapred.cc:306-308 */
  M355_IBF_PCM                     = 1 << 2   /* raw block (slice.cc:4211-4255): res_ofs indexes pcm[] */
};
typedef struct m355_ib {
  uint16_t x, y;                   /* position in component samples */
  uint8_t  cidx;
  uint8_t  log2_size;              /* 2..5 (PCM: up to 5) */
  uint8_t  mode;                   /* IntraPredMode 0..34 */
  uint8_t  flags;                  /* M355_IBF_* */
  uint32_t res_ofs;                /* residual buffer offset (int16 units) / pcm[] offset */
} m355_ib;                         /* 12 bytes */

/* One picture's complete work description. All pointers are HOST pointers owned by the caller;
 * m355_submit_picture() copies what it needs before returning. Lists:
 *   coeffs[i] = (uint16 pos) | (int16 level << 16)  with pos = xC + yC*nT (slice.cc:3441-3447)
 *   rbs[] is grouped by log2_size: rb_count[s] entries of size (s+2), concatenated 4x4,8x8,16,32.
 *   scaling_factors: [sizeId 0..3][matrixID 0..5][y][x] uint8 (sps.h:61-64), 6*(16+64+256+1024) B. */
typedef struct m355_picture {
  m355_pic_params pp;
  int32_t dst_frame;                         /* frame handle receiving the decoded picture  */
  int32_t ref_frames[M355_MAX_REF_FRAMES];   /* frame handles, indexed by m355_pb.ref_slot  */
  int32_t n_slices, n_ctbs, n_cus, n_tus, n_pbs, n_wts, n_ibs;
  int32_t rb_count[4];
  uint32_t n_coeffs, n_pcm, res_len;
  const m355_slice* slices;
  const m355_ctb*   ctbs;                    /* raster order, PicSizeInCtbsY entries */
  const m355_cu*    cus;
  const m355_tu*    tus;
  const m355_pb*    pbs;
  const m355_wt*    wts;
  const m355_rb*    rbs;
  const m355_ib*    ibs;
  const uint32_t*   coeffs;
  const uint16_t*   pcm;
  const uint8_t*    scaling_factors;         /* NULL unless M355_PF_SCALING_LIST */
} m355_picture;

/* ------------------------------------------------------------------------------------------------
 * Context, frames (device-resident DPB), submission.
 * ---------------------------------------------------------------------------------------------- */
typedef struct m355_ctx m355_ctx;

M355_API int  m355_create(int device, m355_ctx** out);
M355_API void m355_destroy(m355_ctx* ctx);

/* Frames: planes of uint8 (bit depth 8) or uint16 (9..16), exactly the reference's plane element
 * types (image.h:295-301). Returns handle >= 0 or a negative M355_ERR_*. */
M355_API int m355_frame_create(m355_ctx* ctx, int width, int height, int chroma_format_idc,
                               int bit_depth_luma, int bit_depth_chroma);
M355_API int m355_frame_destroy(m355_ctx* ctx, int frame);
/* stride in SAMPLES, as everywhere in the reference (acceleration.h: "Strides are in samples").  Both calls block until the plane has arrived; src / dst may be
 * any host memory: the library stages the plane in a pinned buffer of the context and queues the copy on one of its own streams (the frame's last writer's for a
 * download) — a blocking hipMemcpy2D on pageable memory was measured to deliver stale rows when many processes share the GPU. */
M355_API int m355_frame_upload(m355_ctx* ctx, int frame, int cidx, const void* src, ptrdiff_t stride);
M355_API int m355_frame_download(m355_ctx* ctx, int frame, int cidx, void* dst, ptrdiff_t stride);
/* The same for all planes of the frame, asynchronous (picture output, de265_get_next_picture / de265_get_image_plane: image.h's planes
 * are the destination): the copies are queued right behind the decode that writes the frame, on that lane's stream — beside the
 * host and the other lanes' decodes; a later picture decoded into the frame waits for them.  dst[c] / stride[c] (in samples) per plane
 * (unused planes of a monochrome frame are ignored); use m355_host_alloc'ed planes.  m355_frame_download_wait blocks until THIS
 * frame's copy has landed (m355_wait also waits for every copy). */
M355_API int m355_frame_download_async(m355_ctx* ctx, int frame, void* const dst[3], const ptrdiff_t stride[3]);
M355_API int m355_frame_download_wait(m355_ctx* ctx, int frame);
M355_API int m355_frame_fill(m355_ctx* ctx, int frame, int value_luma, int value_chroma);

/* Diagnostic (bench.py's roofline): the device-to-device copy rate this box reaches — `bytes` read + `bytes` written per launch of a float4 copy
 * kernel, the median of `iters` launches timed with events on the context's stream, in GB/s (read + written bytes / time).  The achievable
 * ceiling next to the 8 TB/s specification ("measure the real ceiling with a device-to-device copy kernel on the box"). */
M355_API int m355_measure_copy_rate(m355_ctx* ctx, size_t bytes, int iters, double* gbps);

/* Pinned host memory for the planes an application reads decoded pictures from: what a get_buffer callback registered
 * with de265_set_image_allocation_functions (de265.h:350-368; default allocator image.cc:110-184) hands out, so that
 * m355_frame_download runs at full PCIe rate.  NULL on failure. */
M355_API void* m355_host_alloc(size_t bytes);
M355_API void  m355_host_free(void* p);

/* SEI decoded picture hash of a device frame, computed where the frame lives.
 * Replaces compute_MD5 / compute_CRC_8bit_fast / compute_checksum (sei.cc:161-258) as called by
 * process_sei_decoded_picture_hash (sei.cc:276-356); hash_type and the result fields are those of
 * sei_decoded_picture_hash (sei.h:57-70).  Only the field of the requested type is written, for plane 0 (monochrome)
 * or planes 0..2.  Synchronous: waits for the pictures in flight.  CRC and checksum run on the device (the frame is not
 * copied back); MD5 is one serial chain per plane, so the planes are downloaded and hashed on host threads. */
#define M355_HASH_MD5      0
#define M355_HASH_CRC      1
#define M355_HASH_CHECKSUM 2
typedef struct m355_picture_hash {
  uint8_t  md5[3][16];
  uint16_t crc[3];
  uint32_t checksum[3];
} m355_picture_hash;
M355_API int m355_frame_hash(m355_ctx* ctx, int frame, int hash_type, m355_picture_hash* out);

/* Replaces (deferred): decode_TU (slice.cc:3460), decode_prediction_unit (motion.cc:2190) and
 * run_postprocessing_filters_sequential/_parallel (decctx.cc:1783/1811) for one picture. Asynchronous:
 * returns after the work is enqueued on the context's stream. */
M355_API int m355_submit_picture(m355_ctx* ctx, const m355_picture* pic);
/* Lists recorded IN PLACE: m355_arena_begin hands out, for the NEXT m355_submit_picture on this context, host pointers into
 * the context's pinned staging arena with room for `caps` entries per list (the caller sizes them from the previous
 * picture, say).  The parser's recorder threads write the lists there; the submitted m355_picture carries exactly those
 * pointers (pic->rbs = the 4x4 bin's region; the four size bins live in their own regions caps->rb_bin[0..3], filled in by
 * the call) and its real counts (<= the capacities): the submit then validates, derives its schedules and starts the
 * host-to-device copy without copying a byte on the host.  It waits for the picture that used this arena last (three arenas
 * rotate — measured: a longer ring gains nothing, the submitting thread is the bound).  dst_frame / ref_frames / pp / counts are the caller's to fill in. */
typedef struct m355_arena_caps {
  int32_t n_slices, n_ctbs, n_cus, n_tus, n_pbs, n_wts, n_ibs;
  int32_t n_rbs[4];
  uint32_t n_coeffs, n_pcm;
  int32_t scaling;                 /* 1: room for the scaling-factor tables */
  m355_rb* rb_bin[4];              /* out: where the residual blocks of each size go */
} m355_arena_caps;
M355_API int m355_arena_begin(m355_ctx* ctx, m355_arena_caps* caps, m355_picture* pic);
/* Blocks until all submitted work finished; returns M355_ERR_TIMEOUT if a device spin bound hit, M355_ERR_INVALID if a picture
 * recorded in place was rejected by the device-side list validation (the message names the picture's serial and the record; the
 * next call reports the next rejected picture, if any). */
M355_API int m355_wait(m355_ctx* ctx);
/* Per-picture outcome of asynchronous submits.  The record checks of lists recorded in place (m355_arena_begin) run on the device,
 * ahead of the picture's kernels: m355_submit_picture has returned M355_OK long before a rejection is known.  Every decode gets
 * a serial (1, 2, ...); m355_decode_status(serial) is non-blocking: M355_ERR_BUSY while the decode runs, M355_OK when it finished,
 * M355_ERR_INVALID when its lists were rejected — none of its kernels acted on them, the destination frame was NOT written
 * (m355_last_error names the record).  Kept for the last 64 decodes (an older serial: M355_ERR_STALE — distinct from a rejection, which M355_ERR_INVALID stays for; a
 * rejection that left the ring unreported is still reported by the next m355_wait).  The caller marks the picture and whatever references it
 * as damaged (the reference does the same bookkeeping with de265_image::integrity, image.h:347). */
M355_API unsigned long long m355_last_serial(m355_ctx* ctx);      /* of the decode the last submit / decode call enqueued */
M355_API int m355_decode_status(m355_ctx* ctx, unsigned long long serial);

/* Resident work lists (benchmarks, replay): upload once, decode many times. */
M355_API int m355_picture_upload(m355_ctx* ctx, const m355_picture* pic);   /* -> handle >= 0 */
M355_API int m355_picture_release(m355_ctx* ctx, int handle);
/* other lists into an uploaded picture's arenas: waits for that handle's last decode only (no allocation when they fit) */
M355_API int m355_picture_replace(m355_ctx* ctx, int handle, const m355_picture* pic);
/* m355_arena_begin for the arenas of a RESIDENT picture: list pointers into the pinned arena of `handle` (-1: a new handle) with room
 * for `caps` entries; returns the handle (>= 0) or -error.  The recorder threads write the lists there and
 * m355_picture_replace(ctx, handle, pic) takes them over without copying a byte on the host (same rules as m355_submit_picture on
 * in-place lists).  This is the in-place path of a TILE-SHARDED context (m355_shard_set / m355_group_*), whose pictures are decoded
 * from handles; there `pp` — the picture's parameters, known before its lists are recorded — sizes the room for the border units
 * of the other ranks behind cus[] / pbs[] (unsharded contexts ignore it; the lists of a sharded picture are checked on the host). */
M355_API int m355_picture_arena_begin(m355_ctx* ctx, int handle, m355_arena_caps* caps, const m355_pic_params* pp, m355_picture* pic);
M355_API int m355_decode_resident(m355_ctx* ctx, int handle);
/* n independent INTRA pictures (n <= pipeline depth, distinct destination frames, one chroma format and sample type) with ONE intra
 * stage: each picture's residuals / border plans and its filters run on a lane of its own as for n m355_decode_resident calls, the
 * CTB wavefronts of all n run interleaved inside one k_intra launch — the way all-intra material (the reference's decode_CTB loop
 * over intra slices, slice.cc:4375 read_coding_tree_unit -> intrapred.cc:279 decode_intra_prediction, one picture at a time) fills
 * the GPU without one hardware queue per picture.  The pictures get consecutive serials (m355_last_serial = the last one's);
 * results are those of n single decodes, bit for bit.  Not stage-timed (m355_timing_collect sees single decodes only). */
M355_API int m355_decode_batch(m355_ctx* ctx, const int* handles, int n);
/* stage mask for m355_set_stages: run only part of the chain (stage-isolated parity, like
 * DE265_DECODER_PARAM_DISABLE_DEBLOCKING / _DISABLE_SAO, de265.h:409-410) */
enum { M355_STAGE_INTER = 1, M355_STAGE_RESIDUAL = 2, M355_STAGE_INTRA = 4, M355_STAGE_DEBLOCK = 8,
       M355_STAGE_SAO = 16, M355_STAGE_ALL = 31 };
M355_API int m355_set_stages(m355_ctx* ctx, int stage_mask);
/* Pictures in flight (1..32): 1 (default) = one after the other on the context's stream; n = consecutive decodes go
 * round n lanes (own streams, working planes, scratch) and overlap wherever the frames they touch allow: a decode
 * waits for the last writer of each reference frame it reads and, right before its own first write, for the last writer
 * and the readers of its destination frame.  (The reference decodes independent pictures concurrently too: frame-parallel
 * image units, decctx.cc:605-630.)  Three is the sweet spot for inter pictures; an all-intra stream (dependency-bound k_intra,
 * a fraction of the GPU per picture) gains up to nine: intra pictures on lanes 3.. run on streams of other priority classes, which
 * the HIP runtime gives hardware queues of their own (M355_LANE_PRIORITIES=0 turns that off). */
M355_API int m355_set_pipeline_depth(m355_ctx* ctx, int depth);

/* ------------------------------------------------------------------------------------------------
 * Tile sharding across GPUs (one process per GPU).  Tiles are independent for
 * prediction (intrapred.h:499-508 stops availability at tile borders) but coupled by the in-loop
 * filters when pps.loop_filter_across_tiles_enabled_flag is set (deblock.cc:191-209, sao.cc:158-163)
 * and by motion vectors that reach into other tiles of the reference pictures.  A sharded context
 * owns the tiles t (tile-scan order) with  t * nranks / n_tiles == rank  and is given work lists
 * that hold ONLY its tiles' CUs / TUs / PBs / residual / intra blocks (slices[] and ctbs[] stay
 * picture-wide: the filters read the neighbours' slice flags).  One picture then runs as five phases
 * with an exchange between them; the exchange buffers are CALLER-owned device memory (so the host
 * layer can hand them to RCCL / torch.distributed), all 32-bit-word arrays in a canonical,
 * rank-independent layout in which every element is produced by exactly one rank and is ZERO on all
 * others — an integer SUM all-reduce (or any owner -> neighbour point-to-point copy) completes them:
 *
 *   phase 0  k_meta, k_inter, k_residual, k_intra on the own tiles;
 *            pack X0 = border-unit metadata (QP_Y, PredMode, pcm/bypass, PBMotion, edge + cbf flags of
 *            every 4x4 unit next to an interior tile boundary) + the 4-luma / 2-chroma-sample column
 *            strips either side of every vertical tile boundary, PRE-deblock        -> exchange X0
 *   phase 1  unpack X0 (foreign units appear as local metadata), deblock vertical edges (a boundary edge
 *            is computed by both owners, each writes only its own side); pack X1 = row strips either
 *            side of every horizontal tile boundary, post-vertical-pass             -> exchange X1
 *   phase 2  unpack X1, deblock horizontal edges; pack X2 = column + row strips, deblocked (the
 *            1-sample ring incl. corners that SAO edge classes read)                -> exchange X2
 *   phase 3  unpack X2, SAO into the destination frame (own tiles); pack X3[rank] = own tiles of the
 *            destination frame                                                      -> all-gather X3
 *   phase 4  unpack X3: the destination frame is complete on every rank (reference for later pictures).
 * ---------------------------------------------------------------------------------------------- */
M355_API int m355_shard_set(m355_ctx* ctx, int rank, int nranks);      /* nranks == 0: sharding off (default) */
/* rank owning tile t (tile-scan order) — pure function, the partition rule above */
M355_API int m355_shard_owner_of_tile(int tile, int n_tiles, int nranks);
/* byte size of exchange buffer `which` (0..3) for a resident picture; X3 = nranks equal slots
 * (this rank's slot is [rank*bytes/nranks, (rank+1)*bytes/nranks)).  Multiples of 4. */
M355_API int64_t m355_shard_xbuf_bytes(m355_ctx* ctx, int handle, int which);
/* run one phase (0..4) of a resident picture; xbuf = the buffer the phase packs into (phases 0..3) —
 * the buffer it unpacks is the one passed to the previous phase, which must still be valid and hold
 * the exchanged contents.  Asynchronous.  With m355_set_pipeline_depth(ctx, n >= 2) consecutive pictures run on n lanes: phase 0
 * takes the next lane, the later phases of a picture follow it there, frame hazards are ordered as for m355_decode_resident —
 * the exchanges and filter phases of one picture overlap the prediction phase of the next.  After every call m355_stream() is
 * the stream of the picture's lane: order the exchange that follows on it.
 * A REFERENCE picture is complete only after phase 4 (phase 3 marks the destination frame written with the own tiles only): issue
 * its phase 4 before phase 0 of any picture that reads it — m355_decode_sharded and libde265_amd/shard.py do, they issue a picture's
 * phases back to back. */
M355_API int m355_decode_phase(m355_ctx* ctx, int handle, int phase, void* xbuf);

/* The whole sharded picture in ONE call: the five phases with the exchanges between them, issued from the library (C++; no
 * interpreter between the launches).  The exchange buffers live in the library; the exchanges go through a small callback table,
 * each called in issue order with the stream of the picture's lane (whatever it enqueues on that stream is ordered between the
 * phase that packed the buffer and the phase that unpacks it):
 *   halo_sum    buf holds this rank's elements of X0 / X1 / X2 (zero elsewhere): make it the SUM over this rank and `peers` — the
 *               ranks that own a tile touching one of this rank's (edge or corner; the only producers of what this rank reads,
 *               deblock.cc:191-209, sao.cc:158-163); scratch = room for one buffer per peer
 *   all_gather  X3: nranks slots of slot_bytes, slot `rank` filled -> all slots
 * m355_shard_rccl_init installs the built-in RCCL implementation (librccl is loaded at that moment; neighbour ncclSend / ncclRecv
 * in one group + one add kernel, ncclAllGather in place): one process per GPU, the unique id travels over the application's own
 * bootstrap channel.  Tests install callbacks over another transport (gloo).  gather = 0: a non-reference picture (no X3). */
typedef struct m355_comm {
  void* user;
  int (*halo_sum)(void* user, void* buf, size_t bytes, const int* peers, int n_peers, void* scratch, void* stream);
  int (*all_gather)(void* user, void* buf, size_t slot_bytes, int rank, int nranks, void* stream);
} m355_comm;
M355_API int m355_shard_set_comm(m355_ctx* ctx, const m355_comm* comm);
M355_API int m355_decode_sharded(m355_ctx* ctx, int handle, int gather);
M355_API int m355_rccl_unique_id(void* out128);                                  /* rank 0: ncclGetUniqueId (128 bytes) */
M355_API int m355_shard_rccl_init(m355_ctx* ctx, const void* id128, int rank, int nranks);   /* m355_shard_set + an RCCL communicator on the context's device */
/* The same exchanges between the rank PROCESSES of one node without a collective library (csrc/runtime_ipc.hip): every rank's exchange buffers are
 * exported with hipIpcGetMemHandle through a POSIX shared-memory segment named after `name` (a job-unique string, the same on every rank; rank 0
 * creates the segment) and mapped by their readers — X0..X2: a rank fetches its neighbours' buffers and adds; X3: a rank copies every other rank's
 * finished tiles straight out of that rank's gather buffer (N - 1 concurrent peer reads, one per xGMI link) — ordered by interprocess events
 * (M355_IPC_HOST_SYNC=1: by draining the recording stream instead) and per-rank sequence words in the segment.  Every rank must decode the same
 * pictures in the same order with the same handle numbers (at most 32 handles with exchange buffers, 16 ranks).  m355_shard_set + the callbacks
 * of m355_decode_sharded; needs HSA_ENABLE_IPC_MODE_LEGACY=0 where the driver only has dmabuf IPC.  M355_IPC_TIMEOUT=<seconds> (default 30)
 * bounds every wait for another rank. */
M355_API int m355_shard_ipc_init(m355_ctx* ctx, const char* name, int rank, int nranks);
M355_API int m355_shard_ipc_close(m355_ctx* ctx);
/* collective self-test of that transport: `words` 32-bit words through the halo exchange (every other rank as peer; a lone rank
 * sends to itself) and the all-gather, verified on the host */
M355_API int m355_shard_rccl_selftest(m355_ctx* ctx, size_t words);
/* Tile sharding inside ONE process (no collective library): a group of contexts — one per device, or several on one device —
 * decodes one picture; rank r = position in `ctxs`.  Each context gets its share of the picture with m355_picture_upload (after
 * m355_group_create, which calls m355_shard_set(ctx, r, n)); m355_group_decode(handles[r]) issues every rank's phases and moves
 * the halo buffers / finished tiles between the contexts with hipMemcpyPeerAsync ordered by events on their own streams.
 * Asynchronous like m355_decode_sharded; m355_wait on every context to finish.  This is what a decoder that parses one
 * bitstream with a thread per tile uses to spread the tiles over the GPUs of a node (deblock.cc:191-209, sao.cc:158-163 are
 * the couplings the exchanges carry). */
typedef struct m355_group m355_group;
M355_API int m355_group_create(m355_ctx* const* ctxs, int n, m355_group** out);
M355_API void m355_group_destroy(m355_group* group);
M355_API int m355_group_decode(m355_group* group, const int* handles, int gather);
/* device time (ms) of exchange `which` (0..3) of a sharded picture's buffers over the installed transport, averaged over `iters` runs;
 * collective: every rank calls it alike, after at least one m355_decode_sharded of the picture */
M355_API int m355_shard_time_exchange(m355_ctx* ctx, int handle, int which, int iters, float* ms_each);
/* the ranks `rank` exchanges halos with for these picture parameters (-> count, peers[] filled up to max_peers) */
M355_API int m355_shard_peers(const m355_pic_params* pp, int rank, int nranks, int* peers, int max_peers);

/* Per-stage device timing from HIP events recorded on the context's OWN stream around every decode
 * enqueued between m355_timing_reset() and m355_timing_collect(): averages in milliseconds, stage order
 * [meta, inter, residual, intra, deblock, sao]. m355_timing_collect() waits for the work and ends the window: decodes outside a
 * window record no stage events (seven event packets per picture are a diagnostic, not part of the decode). */
M355_API int m355_timing_reset(m355_ctx* ctx);
M355_API int m355_timing_collect(m355_ctx* ctx, int* n_decodes, float* total_ms, float stage_ms[6]);
M355_API void* m355_stream(m355_ctx* ctx);   /* hipStream_t of the context's active lane (the lane of the last decode / phase call) */

#ifdef __cplusplus
}
#endif
#endif /* DE265_MI355X_H */
