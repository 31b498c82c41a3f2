/*
 * oracle/ref_streamgen.cc — TEST INFRASTRUCTURE ONLY (oracle/_ref/streamgen; never part of the product library).
 *
 * A synthetic HEVC bitstream writer: there are no 4K / 8K tiled 10-bit streams offline and the reference's
 * own encoder only produces 8-bit intra pictures (oracle/ref_encode.cc), so this writes random — but syntactically valid —
 * streams with the reference's OWN bitstream machinery, compiled from /root/reference where it lies:
 *     CABAC_encoder_bitstream                         cabac.h:105     (arithmetic coder, emulation prevention)
 *     video/seq/pic_parameter_set::write              vps.cc:239, sps.cc:1131, pps.cc:761
 *     slice_segment_header::write                     slice.cc:910    (RPS by SPS index, entry points)
 *     encode_residual                                 encoder/encoder-syntax.cc:732 (residual_coding, the complete one)
 *     context_model_table::init, check_CTB_available, fillIntraPredModeCandidates, get_intra_scan_idx (the decoder's own)
 * What is written here is the coding-tree syntax above the residuals (7.3.8.2 - 7.3.8.9: SAO, coding quadtree, coding unit,
 * prediction unit, transform tree), mirroring — function by function — what the reference PARSER reads (slice.cc:1565-2700,
 * 2735-2900, 3870-4700), with a de265_image as the neighbour state so that every context index and inferred value comes
 * from the same helper functions the decoder uses.
 *
 * Streams: picture 0 = IDR (I slice), then P or B pictures referencing the one or two pictures before (RPS in the SPS);
 * uniform tiles (each its own CABAC substream, entry points in the slice header); 8..12 bit; CTB 64, CUs 8..64, all inter
 * partitions incl. AMP, merge / skip / AMVP with random mvd, intra NxN, transform trees of depth <= 2, random sparse
 * coefficient blocks, SAO parameters per CTB, deblocking on.
 *
 * With `features` = 0, chroma 4:2:0 and one slice per picture the headers come from the reference's own writers.  Everything
 * those cannot express is written by OUR writers below, from the syntax tables the reference's PARSER implements (cited per
 * function) — the reference's writers assert or stop there (slice.cc:1176-1178 pred_weight_table, sps.cc:1085-1092
 * scaling_list_data, pps.cc:880-891 / sps.cc:1302-1308 range extensions, encoder-syntax.cc:745 transform_skip_flag):
 *   F_WP        explicit weighted prediction (pred_weight_table, slice.cc:159-231)
 *   F_TSKIP     transform_skip_flag (slice.cc:2963-2970); with F_REXT also on blocks up to 32x32 (pps range extension)
 *   F_BYPASS    cu_transquant_bypass_flag (slice.cc:4345-4356)
 *   F_QPDELTA   cu_qp_delta_abs / sign (slice.cc:3618-3650, quantisation groups slice.cc:4671-4679), slice_qp_delta
 *   F_PCM       pcm_flag + pcm_sample (slice.cc:4420-4434, 4211-4283), PCM bit depths below the sample bit depth
 *   F_SCALING   scaling lists: the SPS default lists; F_SCALING_PPS: explicit lists in the PPS (scaling_list_data, sps.cc:939-1081)
 *   F_REXT      range extensions: implicit / explicit RDPCM (slice.cc:2974-2985), transform_skip_rotation, cross-component
 *               prediction in 4:4:4 (slice.cc:3527-3583, 3710-3760), cu_chroma_qp_offset (slice.cc:3655-3683)
 *   F_CIP       constrained_intra_pred_flag
 *   F_DEPSLICE  every other slice segment of a picture is a dependent one (slice.cc:503-520)
 *   chroma      0 monochrome, 1 4:2:0, 2 4:2:2, 3 4:4:4 (transform unit layout slice.cc:3584-3860, chroma cbf pairs
 *               slice.cc:3940-3960, chroma prediction modes slice.cc:4537-4575)
 *   slices      slice segments per picture, each with its own deblocking override (disable flag, beta / tc offsets),
 *               SAO flags, loop-filter-across-slices flag and QP (slice.cc:750-826)
 *   F_RA        a random-access stream shape (the offline stand-in for the ra_main conformance streams): hierarchical-B groups of
 *               8 pictures coded in the order 8 4 2 1 3 6 5 7 (output order != decoding order, sps_max_num_reorder_pics 3:
 *               decctx.cc:1885-2035, dpb.cc:194-281), every picture's reference picture set written in its slice header
 *               (st_ref_pic_set, refpic.cc:85-260: pictures before AND after the current one, some kept but not used), four
 *               active references per list out of 1..6 pictures (cyclic list construction, decctx.cc:1550-1700), odd pictures
 *               are sub-layer non-reference pictures (TRAIL_N)
 *   F_LT        (with F_RA) the IDR picture becomes a long-term reference from the third group on (slice.cc:517-598, lt_idx /
 *               poc_lsb_lt / used_by_curr_pic_lt_flag in the slice header)
 *   F_TMVP      sps_temporal_mvp_enabled_flag: slice_temporal_mvp_enabled_flag, collocated_from_l0_flag, collocated_ref_idx
 *               (slice.cc:705-728) — the decoder derives the temporal merge / AMVP candidates (motion.cc:1210-1560)
 *   F_SDH       sign_data_hiding_enabled_flag (slice.cc:3317-3440; the reference's residual writer implements the hidden sign)
 *   F_WPP       entropy_coding_sync_enabled_flag: one substream per CTB row with the context models of the row above's second
 *               CTB (decode_substream, slice.cc:4732-4900; what decode_slice_unit_WPP threads parse, decctx.cc:840-1061)
 *
 *   F_MIXSLICE  every slice of a picture its own slice type (I / P slices inside P / B pictures), number of active references,
 *               MaxNumMergeCand and cabac_init_flag (slice.cc:639-702, 1280-1282): the per-slice tables of the decoder and of
 *               whoever records its decisions
 *   F_LISTMOD   ref_pic_lists_modification (slice.cc:668-690, list construction decctx.cc:1580-1640): lists that name the same
 *               picture twice, or in another order than the default one
 *   F_NOOUTPUT  pic_output_flag = 0 on some pictures (slice.cc:469-475): decoded, referenced, never shown (decctx.cc:1851, 1974)
 *   geom        picture / block geometry other than the default (CTB 64, CBs 8..64, TBs 4..32, uniform tiles): bits 0-1 CTB 32 (1) or
 *               16 (2); G_MINCB16 coding blocks of at least 16x16 (no 8x8 CUs, so no intra NxN below 8x8 PUs and no 8x4 / 4x8 PBs);
 *               G_TILES uniform_spacing_flag = 0 with explicit column widths / row heights (pps.cc:392-421); G_NOTILEFILTER
 *               loop_filter_across_tiles_enabled_flag = 0 (deblock.cc:191-209, sao.cc:158-163); G_PARMERGE
 *               log2_parallel_merge_level = 4 (shared merge candidates, motion.cc:1630-1680); G_TB16 TBs 4..16 and
 *               strong_intra_smoothing off; G_CONFWIN a conformance window (sps.cc:228-262: what the application sees is
 *               a cropped picture, image.cc:434-470 pixels_confwin)
 *
 * usage: streamgen out.h265 W H bit_depth tile_cols tile_rows n_frames seed [intra_pct=5] [b_frames=1] [sao=1]
 *                  [features=0] [chroma=1] [slices=1] [geom=0]
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <memory>
#include <vector>

#include "libde265/de265.h"
#include "libde265/cabac.h"
#include "libde265/contextmodel.h"
#include "libde265/decctx.h"
#include "libde265/image.h"
#include "libde265/intrapred.h"
#include "libde265/nal.h"
#include "libde265/pps.h"
#include "libde265/slice.h"
#include "libde265/sps.h"
#include "libde265/vps.h"
#include "libde265/encoder/encoder-context.h"
#include "libde265/encoder/encoder-syntax.h"
#include "libde265/encoder/encoder-types.h"

/* defined (not static) in encoder/encoder-syntax.cc:732 / :1362, but not declared in its header */
void encode_residual(encoder_context* ectx, CABAC_encoder* cabac, const enc_tb* tb, const enc_cb* cb, int x0, int y0, int log2TrafoSize, int cIdx);
void encode_mvd(encoder_context* ectx, CABAC_encoder* cabac, const int16_t mvd[2]);

namespace {

enum { F_WP = 1, F_TSKIP = 2, F_BYPASS = 4, F_QPDELTA = 8, F_PCM = 16, F_SCALING = 32, F_SCALING_PPS = 64, F_REXT = 256, F_CIP = 512, F_DEPSLICE = 1024,
       F_RA = 2048, F_WPP = 4096, F_TMVP = 8192, F_SDH = 16384, F_LT = 32768, F_MIXSLICE = 65536, F_LISTMOD = 131072, F_NOOUTPUT = 262144 };
enum { G_CTB = 3, G_MINCB16 = 4, G_TILES = 8, G_NOTILEFILTER = 16, G_PARMERGE = 32, G_TB16 = 64, G_CONFWIN = 128 };
struct Cfg { int W, H, bd, tc, tr, frames; uint32_t seed; int intra_pct, b_frames, sao, features, chroma, slices, geom; };

struct Gen {
  Cfg cfg;
  std::shared_ptr<video_parameter_set> vps;
  std::shared_ptr<seq_parameter_set> sps;
  std::shared_ptr<pic_parameter_set> pps;
  decoder_context dctx;                       /* owner of the state image's allocation functions */
  encoder_context ectx;                       /* what encode_residual wants to see (img -> sps / pps) */
  de265_image img;                            /* neighbour state: exactly the metadata the decoder keeps while parsing */
  slice_segment_header* shdr = nullptr;
  context_model_table ctx;
  CABAC_encoder_bitstream* cabac = nullptr;
  uint32_t s = 1;
  int slice_type = SLICE_TYPE_I, nref[2] = {0, 0};
  int slice_index = 0;                        /* index of shdr in img.slices */
  bool qg_coded = true, cqo_coded = true;     /* IsCuQpDeltaCoded / IsCuChromaQpOffsetCoded of the parser (slice.cc:4671-4685) */
  bool cu_bypass = false;                     /* cu_transquant_bypass_flag of the coding unit being written */
  /* the picture being written (F_RA / F_LT / F_TMVP): its reference picture set as the slice header codes it */
  std::vector<int> rps_neg, rps_neg_used, rps_pos, rps_pos_used;   /* POC distances (> 0), nearest first */
  std::vector<int> lt_lsb, lt_used;
  int col_from_l0 = 1, col_ref_idx = 0;
  int list_total = 0, list_bits_total = 0;    /* NumPocTotalCurr: the pictures a modified list may name, and what sizes list_entry_lX (slice.cc:665-668) */

  uint32_t rnd() { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return s; }
  int below(int n) { return (int)(rnd() % (uint32_t)n); }
  int range(int lo, int hi) { return lo + below(hi - lo + 1); }
  bool pct(int p) { return below(100) < p; }
  bool has(int f) const { return (cfg.features & f) != 0; }
  int cat() const { return sps->ChromaArrayType; }

  void bit(int model, int b) { cabac->write_CABAC_bit(model, b); }
  void bypass(int b) { cabac->write_CABAC_bypass(b); }

  /* ---------------- 7.3.8.3 sample adaptive offset (mirrors read_sao, slice.cc:2735-2871) ---------------- */
  void write_sao(int xCtb, int yCtb, int ctbAddrRS)
  {
    const seq_parameter_set& S = *sps;
    const int W = S.PicWidthInCtbsY;
    int merge_left = 0, merge_up = 0;
    if (xCtb > 0) {
      const bool inSlice = ctbAddrRS > (int)shdr->SliceAddrRS;
      const bool inTile = pps->scan->TileIdRS[xCtb + yCtb * W] == pps->scan->TileIdRS[xCtb - 1 + yCtb * W];
      if (inSlice && inTile) { merge_left = pct(20); bit(CONTEXT_MODEL_SAO_MERGE_FLAG, merge_left); }
    }
    if (yCtb > 0 && !merge_left) {
      const bool inSlice = (ctbAddrRS - W) >= (int)shdr->SliceAddrRS;
      const bool inTile = pps->scan->TileIdRS[xCtb + yCtb * W] == pps->scan->TileIdRS[xCtb + (yCtb - 1) * W];
      if (inSlice && inTile) { merge_up = pct(20); bit(CONTEXT_MODEL_SAO_MERGE_FLAG, merge_up); }
    }
    if (!merge_left && !merge_up) {
      int type_c = 0;
      const int nc = cat() ? 3 : 1;
      for (int c = 0; c < nc; c++) {
        if (!((shdr->slice_sao_luma_flag && c == 0) || (shdr->slice_sao_chroma_flag && c > 0))) continue;
        int type;
        if (c < 2) {
          type = below(3);                                                     /* 0 off, 1 band, 2 edge */
          bit(CONTEXT_MODEL_SAO_TYPE_IDX, type != 0);
          if (type) bypass(type == 2);
          if (c == 1) type_c = type;
        } else type = type_c;
        if (!type) continue;
        const int bitDepth = c ? S.BitDepth_C : S.BitDepth_Y;
        const int cMax = (1 << (std::min(bitDepth, 10) - 5)) - 1;
        int off[4];
        for (int i = 0; i < 4; i++) { off[i] = below(std::min(cMax, 7) + 1); cabac->write_CABAC_TU_bypass(off[i], cMax); }
        if (type == 1) {
          for (int i = 0; i < 4; i++) if (off[i]) bypass(below(2));
          cabac->write_CABAC_FL_bypass(below(32), 5);                          /* band position */
        } else if (c < 2) cabac->write_CABAC_FL_bypass(below(4), 2);           /* edge class (chroma: shared by Cb / Cr) */
      }
    }
  }

  /* ---------------- 7.3.8.4 coding quadtree (read_coding_quadtree, slice.cc:4648-4715) ---------------- */
  void coding_quadtree(int x0, int y0, int log2, int depth, int target)
  {
    const seq_parameter_set& S = *sps;
    int split;
    if (x0 + (1 << log2) <= S.pic_width_in_luma_samples && y0 + (1 << log2) <= S.pic_height_in_luma_samples && log2 > S.Log2MinCbSizeY) {
      split = log2 > target;
      const int availL = check_CTB_available(&img, x0, y0, x0 - 1, y0), availA = check_CTB_available(&img, x0, y0, x0, y0 - 1);
      const int condL = availL && img.get_ctDepth(x0 - 1, y0) > depth, condA = availA && img.get_ctDepth(x0, y0 - 1) > depth;
      bit(CONTEXT_MODEL_SPLIT_CU_FLAG + condL + condA, split);
    } else split = log2 > S.Log2MinCbSizeY;
    /* quantisation groups / chroma QP offset groups start here (slice.cc:4671-4685) */
    if (pps->cu_qp_delta_enabled_flag && log2 >= pps->Log2MinCuQpDeltaSize) qg_coded = false;
    if (shdr->cu_chroma_qp_offset_enabled_flag && log2 >= pps->Log2MinCuChromaQpOffsetSize) cqo_coded = false;
    if (split) {
      const int x1 = x0 + (1 << (log2 - 1)), y1 = y0 + (1 << (log2 - 1));
      coding_quadtree(x0, y0, log2 - 1, depth + 1, target);
      if (x1 < S.pic_width_in_luma_samples) coding_quadtree(x1, y0, log2 - 1, depth + 1, target);
      if (y1 < S.pic_height_in_luma_samples) coding_quadtree(x0, y1, log2 - 1, depth + 1, target);
      if (x1 < S.pic_width_in_luma_samples && y1 < S.pic_height_in_luma_samples) coding_quadtree(x1, y1, log2 - 1, depth + 1, target);
    } else {
      img.set_ctDepth(x0, y0, log2, depth);
      coding_unit(x0, y0, log2, depth);
    }
  }

  /* ---------------- 7.3.8.6 prediction unit (read_prediction_unit, slice.cc:4122-4205) ---------------- */
  void write_merge_idx(int idx)
  {
    const int maxc = shdr->MaxNumMergeCand;
    if (maxc <= 1) return;
    bit(CONTEXT_MODEL_MERGE_IDX, idx ? 1 : 0);
    if (idx > 0) {
      int i = 1;
      while (i < maxc - 1) { const int inc = i < idx; bypass(inc); if (inc) i++; else break; }
    }
  }
  void write_ref_idx(int idx, int nActive)
  {
    if (nActive <= 1) return;
    const int cMax = nActive - 1;
    for (int i = 0; i < cMax; i++) {
      const int b = idx > i;
      if (i == 0) bit(CONTEXT_MODEL_REF_IDX_LX + 0, b); else if (i == 1) bit(CONTEXT_MODEL_REF_IDX_LX + 1, b); else bypass(b);
      if (!b) break;
    }
  }
  void write_mvd()
  {
    int16_t mvd[2];
    for (int k = 0; k < 2; k++) mvd[k] = (int16_t)(pct(30) ? 0 : (pct(85) ? range(-8, 8) : range(-96, 96)));
    encode_mvd(&ectx, cabac, mvd);                                             /* encoder-syntax.cc:1362 (needs the coder only) */
  }
  bool prediction_unit(int nPbW, int nPbH, int ctDepth)                        /* returns merge_flag */
  {
    const int merge = pct(45);
    bit(CONTEXT_MODEL_MERGE_FLAG, merge);
    if (merge) { write_merge_idx(below(shdr->MaxNumMergeCand)); return true; }
    int idc = 0;                                                               /* 0 L0, 1 L1, 2 BI */
    if (slice_type == SLICE_TYPE_B) {
      idc = nPbW + nPbH == 12 ? below(2) : below(3);
      if (nPbW + nPbH == 12) bit(CONTEXT_MODEL_INTER_PRED_IDC + 4, idc);
      else {
        bit(CONTEXT_MODEL_INTER_PRED_IDC + ctDepth, idc == 2);
        if (idc != 2) bit(CONTEXT_MODEL_INTER_PRED_IDC + 4, idc);
      }
    }
    if (idc != 1) { write_ref_idx(below(nref[0]), nref[0]); write_mvd(); bit(CONTEXT_MODEL_MVP_LX_FLAG, below(2)); }
    if (idc != 0) { write_ref_idx(below(nref[1]), nref[1]); write_mvd(); bit(CONTEXT_MODEL_MVP_LX_FLAG, below(2)); }
    return false;
  }

  /* ---------------- 7.3.8.11 residual_coding: what stands in front of the coefficients (slice.cc:2963-2985) is written here
     (transform_skip_flag, explicit_rdpcm_flag / _dir), the coefficients by the reference's own writer; (x0, y0) = the luma
     position the parser looks the prediction modes up at ---------------- */
  void residual(int x0, int y0, int log2, int cIdx, bool intra)
  {
    const int n = 1 << log2;
    bool tskip = false;
    if (pps->transform_skip_enabled_flag && !cu_bypass && log2 <= pps->Log2MaxTransformSkipSize) {
      tskip = pct(45);
      bit(CONTEXT_MODEL_TRANSFORM_SKIP_FLAG + (cIdx ? 1 : 0), tskip);
    }
    if (!intra && sps->range_extension.explicit_rdpcm_enabled_flag && (tskip || cu_bypass)) {
      const int f = pct(60);
      bit(CONTEXT_MODEL_RDPCM_FLAG + (cIdx ? 1 : 0), f);
      if (f) bit(CONTEXT_MODEL_RDPCM_DIR + (cIdx ? 1 : 0), below(2));
    }
    enc_cb cb;
    cb.PredMode = intra ? MODE_INTRA : MODE_INTER;
    enc_tb tb(x0, y0, log2, &cb);
    tb.alloc_coeff_memory(cIdx, n);
    int16_t* c = tb.coeff[cIdx];
    memset(c, 0, sizeof(int16_t) * n * n);
    const int r = below(100);
    const int nnz = r < 70 ? 1 + below(std::max(1, n * n / 16)) : (r < 95 ? 1 + below(std::max(1, n * n / 4)) : n * n);
    const int lim = below(4) ? std::max(2, n / 2) : n;                          /* low-frequency biased */
    for (int i = 0; i < nnz; i++) {
      const int px = below(lim), py = below(lim);
      int v = r < 95 ? range(-24, 24) : range(-400, 400);
      if (v == 0) v = 1;
      c[px + py * n] = (int16_t)v;
    }
    if (intra) {
      tb.intra_mode = (enum IntraPredMode)img.get_IntraPredMode(x0, y0);
      tb.intra_mode_chroma = (enum IntraPredMode)img.get_IntraPredModeC(x0, y0);
    }
    if (pps->sign_data_hiding_flag) {
      /* the residual writer leaves out the sign of a sub-block's first coefficient in scan order when it is hidden and asserts
         that it was positive (encoder-syntax.cc:1074-1090); the decoder infers it from the parity of the level sum
         (slice.cc:3426-3440) — whichever of the three scans applies, the first coefficient is made positive here */
      cb.cu_transquant_bypass_flag = cu_bypass;
      for (int sy = 0; sy < n; sy += 4)
        for (int sx = 0; sx < n; sx += 4) {
          int first[3] = {-1, -1, -1};
          for (int d = 0; d < 7 && first[0] < 0; d++)                             /* up-right diagonal: anti-diagonals from the top left */
            for (int yy = std::min(d, 3); yy >= 0 && first[0] < 0; yy--) { const int xx = d - yy; if (xx < 4 && c[sx + xx + (sy + yy) * n]) first[0] = sx + xx + (sy + yy) * n; }
          for (int i = 0; i < 16 && first[1] < 0; i++) if (c[sx + (i & 3) + (sy + (i >> 2)) * n]) first[1] = sx + (i & 3) + (sy + (i >> 2)) * n;    /* horizontal */
          for (int i = 0; i < 16 && first[2] < 0; i++) if (c[sx + (i >> 2) + (sy + (i & 3)) * n]) first[2] = sx + (i >> 2) + (sy + (i & 3)) * n;    /* vertical */
          for (int k = 0; k < 3; k++) if (first[k] >= 0 && c[first[k]] < 0) c[first[k]] = (int16_t)-c[first[k]];
        }
    }
    encode_residual(&ectx, cabac, &tb, &cb, x0, y0, log2, cIdx);
  }

  /* cross_comp_pred (slice.cc:3527-3583) */
  void write_cross_comp_pred(int c)
  {
    const int v = pct(35) ? 0 : 1 + below(4);                                   /* log2_res_scale_abs_plus1, TU binarisation, cMax 4 */
    for (int b = 0; b < 4; b++) {
      const int more = v > b;
      bit(CONTEXT_MODEL_LOG2_RES_SCALE_ABS_PLUS1 + 4 * c + b, more);
      if (!more) break;
    }
    if (v) bit(CONTEXT_MODEL_RES_SCALE_SIGN_FLAG + c, below(2));
  }

  /* ---------------- 7.3.8.10 transform unit (read_transform_unit, slice.cc:3584-3860) ---------------- */
  void transform_unit(int x0, int y0, int xBase, int yBase, int log2, int blkIdx, bool intra, int cbf_luma, int cbf_cb, int cbf_cr)
  {
    const int CAT = cat();
    const int log2C = std::max(2, CAT == CHROMA_444 ? log2 : log2 - 1);
    const int cbfChroma = cbf_cb | cbf_cr;
    if (cbf_luma || cbfChroma) {
      if (pps->cu_qp_delta_enabled_flag && !qg_coded) {
        /* cu_qp_delta_abs: prefix TU (cMax 5, first bin its own context), suffix EG0; CuQpDeltaVal stays far inside its range */
        const int a = pct(40) ? 0 : (pct(90) ? range(1, 4) : range(5, 9));
        bit(CONTEXT_MODEL_CU_QP_DELTA_ABS + 0, a > 0);
        if (a > 0) {
          for (int i = 1; i < 5; i++) { const int more = a > i; bit(CONTEXT_MODEL_CU_QP_DELTA_ABS + 1, more); if (!more) break; }
          if (a >= 5) cabac->write_CABAC_EGk(a - 5, 0);
          bypass(below(2));                                                     /* cu_qp_delta_sign_flag */
        }
        qg_coded = true;
      }
      if (shdr->cu_chroma_qp_offset_enabled_flag && cbfChroma && !cu_bypass && !cqo_coded) {
        const int f = pct(60);
        bit(CONTEXT_MODEL_CU_CHROMA_QP_OFFSET_FLAG, f);
        if (f && pps->range_extension.chroma_qp_offset_list_len > 1) bit(CONTEXT_MODEL_CU_CHROMA_QP_OFFSET_IDX, below(2));
        cqo_coded = true;
      }
    }
    if (cbf_luma) residual(x0, y0, log2, 0, intra);
    if (log2 > 2 || CAT == CHROMA_444) {
      const bool cross = pps->range_extension.cross_component_prediction_enabled_flag && cbf_luma && (!intra || img.is_IntraPredModeC_Mode4(x0, y0));
      const int yOff = (1 << log2C) * sps->SubHeightC;                          /* 4:2:2: the second chroma block of the unit, in luma rows */
      if (cross) write_cross_comp_pred(0);
      if (cbf_cb & 1) residual(x0, y0, log2C, 1, intra);
      if (CAT == CHROMA_422 && (cbf_cb & 2)) residual(x0, y0 + yOff, log2C, 1, intra);
      if (cross) write_cross_comp_pred(1);
      if (cbf_cr & 1) residual(x0, y0, log2C, 2, intra);
      if (CAT == CHROMA_422 && (cbf_cr & 2)) residual(x0, y0 + yOff, log2C, 2, intra);
    } else if (blkIdx == 3) {
      if (cbf_cb & 1) residual(xBase, yBase, 2, 1, intra);
      if (cbf_cb & 2) residual(xBase, yBase + 4, 2, 1, intra);
      if (cbf_cr & 1) residual(xBase, yBase, 2, 2, intra);
      if (cbf_cr & 2) residual(xBase, yBase + 4, 2, 2, intra);
    }
  }

  /* ---------------- 7.3.8.8 transform tree (read_transform_tree, slice.cc:3870-4025) ---------------- */
  void transform_tree(int x0, int y0, int xBase, int yBase, int log2, int depth, int blkIdx, int maxDepth, int intraSplit,
                      bool intra, int partMode, int parent_cb, int parent_cr, int cbf_pct)
  {
    const seq_parameter_set& S = *sps;
    const int CAT = cat();
    int split;
    if (log2 <= S.Log2MaxTrafoSize && log2 > S.Log2MinTrafoSize && depth < maxDepth && !(intraSplit && depth == 0)) {
      split = pct(30);
      bit(CONTEXT_MODEL_SPLIT_TRANSFORM_FLAG + 5 - log2, split);
    } else {
      const int interSplit = S.max_transform_hierarchy_depth_inter == 0 && depth == 0 && !intra && partMode != PART_2Nx2N;
      split = log2 > S.Log2MaxTrafoSize || (intraSplit && depth == 0) || interSplit;
    }
    if (split) img.set_split_transform_flag(x0, y0, depth);
    int cbf_cb = -1, cbf_cr = -1;
    if ((log2 > 2 && CAT != CHROMA_MONO) || CAT == CHROMA_444) {
      const bool two = CAT == CHROMA_422 && (!split || log2 == 3);             /* 4:2:2: a flag per stacked chroma block (slice.cc:3940-3960) */
      if (parent_cb) { cbf_cb = pct(cbf_pct); bit(CONTEXT_MODEL_CBF_CHROMA + depth, cbf_cb); if (two) { const int b = pct(cbf_pct); bit(CONTEXT_MODEL_CBF_CHROMA + depth, b); cbf_cb |= b << 1; } }
      if (parent_cr) { cbf_cr = pct(cbf_pct); bit(CONTEXT_MODEL_CBF_CHROMA + depth, cbf_cr); if (two) { const int b = pct(cbf_pct); bit(CONTEXT_MODEL_CBF_CHROMA + depth, b); cbf_cr |= b << 1; } }
    }
    if (cbf_cb < 0) cbf_cb = (depth > 0 && log2 == 2) ? parent_cb : 0;
    if (cbf_cr < 0) cbf_cr = (depth > 0 && log2 == 2) ? parent_cr : 0;
    if (split) {
      const int x1 = x0 + (1 << (log2 - 1)), y1 = y0 + (1 << (log2 - 1));
      transform_tree(x0, y0, x0, y0, log2 - 1, depth + 1, 0, maxDepth, intraSplit, intra, partMode, cbf_cb, cbf_cr, cbf_pct);
      transform_tree(x1, y0, x0, y0, log2 - 1, depth + 1, 1, maxDepth, intraSplit, intra, partMode, cbf_cb, cbf_cr, cbf_pct);
      transform_tree(x0, y1, x0, y0, log2 - 1, depth + 1, 2, maxDepth, intraSplit, intra, partMode, cbf_cb, cbf_cr, cbf_pct);
      transform_tree(x1, y1, x0, y0, log2 - 1, depth + 1, 3, maxDepth, intraSplit, intra, partMode, cbf_cb, cbf_cr, cbf_pct);
      return;
    }
    int cbf_luma = 1;
    if (intra || depth != 0 || cbf_cb || cbf_cr) { cbf_luma = pct(cbf_pct); bit(CONTEXT_MODEL_CBF_LUMA + (depth == 0), cbf_luma); }
    transform_unit(x0, y0, xBase, yBase, log2, blkIdx, intra, cbf_luma, cbf_cb, cbf_cr);
  }

  /* ---------------- pcm_sample (read_pcm_samples, slice.cc:4211-4283): after pcm_flag the arithmetic coder is flushed, a one bit
     and alignment zeros follow (9.3.2.5 / 9.3.4.3.5), then the raw samples, then the coder starts afresh ---------------- */
  void pcm_samples(int log2)
  {
    const seq_parameter_set& S = *sps;
    cabac->flush_CABAC();
    cabac->add_trailing_bits();
    const int nc = cat() ? 3 : 1;
    for (int c = 0; c < nc; c++) {
      const int w = (1 << log2) / (c ? S.SubWidthC : 1), h = (1 << log2) / (c ? S.SubHeightC : 1);
      const int nb = c ? S.pcm_sample_bit_depth_chroma : S.pcm_sample_bit_depth_luma;
      const int base = below(1 << nb);
      for (int i = 0; i < w * h; i++) cabac->write_bits((uint32_t)((base + range(-3, 3)) & ((1 << nb) - 1)), nb);
    }
    cabac->flush_VLC();
    cabac->init_CABAC();
  }

  /* ---------------- 7.3.8.5 coding unit (read_coding_unit, slice.cc:4315-4636) ---------------- */
  void coding_unit(int x0, int y0, int log2, int ctDepth)
  {
    const seq_parameter_set& S = *sps;
    const int nCbS = 1 << log2;
    img.set_log2CbSize(x0, y0, log2, true);
    img.clear_split_transform_flags(x0, y0, log2);
    cu_bypass = false;
    if (pps->transquant_bypass_enable_flag) {
      cu_bypass = pct(18);
      bit(CONTEXT_MODEL_CU_TRANSQUANT_BYPASS_FLAG, cu_bypass);
      if (cu_bypass) img.set_cu_transquant_bypass(x0, y0, log2);
    }
    int skip = 0;
    if (slice_type != SLICE_TYPE_I) {
      skip = pct(20);
      const int availL = check_CTB_available(&img, x0, y0, x0 - 1, y0), availA = check_CTB_available(&img, x0, y0, x0, y0 - 1);
      const int condL = availL && img.get_cu_skip_flag(x0 - 1, y0), condA = availA && img.get_cu_skip_flag(x0, y0 - 1);
      bit(CONTEXT_MODEL_CU_SKIP_FLAG + condL + condA, skip);
    }
    if (skip) {     /* (the decoder reads the neighbours' skip flags back as PredMode == MODE_SKIP, image.h:502) */
      write_merge_idx(below(shdr->MaxNumMergeCand));
      img.set_PartMode(x0, y0, PART_2Nx2N);
      img.set_pred_mode(x0, y0, log2, MODE_SKIP);
      return;
    }
    bool intra = true;
    if (slice_type != SLICE_TYPE_I) { intra = pct(cfg.intra_pct); bit(CONTEXT_MODEL_PRED_MODE_FLAG, intra); }
    img.set_pred_mode(x0, y0, log2, intra ? MODE_INTRA : MODE_INTER);
    int part = PART_2Nx2N, intraSplit = 0;
    if (intra) {
      if (log2 == S.Log2MinCbSizeY) {
        part = pct(40) ? PART_NxN : PART_2Nx2N;
        bit(CONTEXT_MODEL_PART_MODE, part == PART_2Nx2N);
        intraSplit = part == PART_NxN;
      }
    } else {
      const int r = below(10);
      const bool amp_ok = S.amp_enabled_flag && log2 > S.Log2MinCbSizeY;
      part = r < 5 ? PART_2Nx2N : (r < 7 ? PART_2NxN : (r < 9 ? PART_Nx2N : (amp_ok ? PART_2NxnU + below(4) : PART_2NxN)));
      const bool nxn_ok = log2 == S.Log2MinCbSizeY && log2 > 3;                  /* four square PBs: only the smallest CB, and not 8x8 */
      if (nxn_ok && r == 9) part = PART_NxN;
      bit(CONTEXT_MODEL_PART_MODE + 0, part == PART_2Nx2N);
      if (part != PART_2Nx2N) {
        const bool horiz = part == PART_2NxN || part == PART_2NxnU || part == PART_2NxnD;
        bit(CONTEXT_MODEL_PART_MODE + 1, horiz);
        if (log2 > S.Log2MinCbSizeY && S.amp_enabled_flag) {
          const bool plain = part == PART_2NxN || part == PART_Nx2N;
          bit(CONTEXT_MODEL_PART_MODE + 3, plain);
          if (!plain) bypass(part == PART_2NxnD || part == PART_nRx2N);
        } else if (nxn_ok && !horiz) bit(CONTEXT_MODEL_PART_MODE + 2, part == PART_Nx2N);
        /* log2 == min: bit1 = 1 -> 2NxN; 0 -> Nx2N at 8x8, one more bin (Nx2N / NxN) above (slice.cc:1773-1791) */
      }
    }
    img.set_PartMode(x0, y0, (enum PartMode)part);
    bool merge_2Nx2N = false;
    if (intra) {
      if (part == PART_2Nx2N && S.pcm_enabled_flag && log2 >= S.Log2MinIpcmCbSizeY && log2 <= S.Log2MaxIpcmCbSizeY) {
        const int pcm = pct(30);
        cabac->write_CABAC_term_bit(pcm);                                        /* pcm_flag (slice.cc:4420-4426) */
        if (pcm) {
          img.set_pcm_flag(x0, y0, log2);
          pcm_samples(log2);
          return;
        }
      }
      /* prev_intra_luma_pred_flag x n, then mpm_idx / rem_intra_luma_pred_mode per block; the resulting modes are derived as
         the decoder derives them (slice.cc:4436-4500) because the residual scan order depends on them */
      const int pbOffset = part == PART_NxN ? nCbS / 2 : nCbS, log2PU = part == PART_NxN ? log2 - 1 : log2;
      int prev[4], mpm[4], rem[4], idx = 0;
      for (int j = 0; j < nCbS; j += pbOffset)
        for (int i = 0; i < nCbS; i += pbOffset) { prev[idx] = pct(60); bit(CONTEXT_MODEL_PREV_INTRA_LUMA_PRED_FLAG, prev[idx]); idx++; }
      const int availA0 = check_CTB_available(&img, x0, y0, x0 - 1, y0), availB0 = check_CTB_available(&img, x0, y0, x0, y0 - 1);
      idx = 0;
      for (int j = 0; j < nCbS; j += pbOffset)
        for (int i = 0; i < nCbS; i += pbOffset) {
          if (prev[idx]) { mpm[idx] = below(3); cabac->write_CABAC_TU_bypass(mpm[idx], 2); }
          else { rem[idx] = below(32); cabac->write_CABAC_FL_bypass(rem[idx], 5); }
          const int x = x0 + i, y = y0 + j;
          const int PUidx = (x >> S.Log2MinPUSize) + (y >> S.Log2MinPUSize) * S.PicWidthInMinPUs;
          enum IntraPredMode cand[3];
          fillIntraPredModeCandidates(cand, x, y, PUidx, availA0 || i > 0, availB0 || j > 0, &img);
          int mode;
          if (prev[idx]) mode = cand[mpm[idx]];
          else {
            if (cand[0] > cand[1]) std::swap(cand[0], cand[1]);
            if (cand[0] > cand[2]) std::swap(cand[0], cand[2]);
            if (cand[1] > cand[2]) std::swap(cand[1], cand[2]);
            mode = rem[idx];
            for (int k = 0; k <= 2; k++) if (mode >= cand[k]) mode++;
          }
          img.set_IntraPredMode(PUidx, log2PU, (enum IntraPredMode)mode);
          idx++;
        }
      /* intra_chroma_pred_mode (slice.cc:4537-4575): one per prediction block in 4:4:4, one per coding unit in 4:2:0 / 4:2:2
         (mapped through Table 8-3 for 4:2:2), none in monochrome; 4 = derived from luma */
      auto chroma_mode = [&](int lumaMode, int icpm) {
        if (icpm == 4) return lumaMode;
        static const int tab[4] = {0, 26, 10, 1};
        return tab[icpm] == lumaMode ? 34 : tab[icpm];
      };
      auto write_icpm = [&]() {
        const int icpm = below(5);
        bit(CONTEXT_MODEL_INTRA_CHROMA_PRED_MODE, icpm != 4);
        if (icpm != 4) cabac->write_CABAC_FL_bypass(icpm, 2);
        return icpm;
      };
      if (cat() == CHROMA_444) {
        for (int j = 0; j < nCbS; j += pbOffset)
          for (int i = 0; i < nCbS; i += pbOffset) {
            const int icpm = write_icpm();
            img.set_IntraPredModeC(x0 + i, y0 + j, log2PU, (enum IntraPredMode)chroma_mode(img.get_IntraPredMode(x0 + i, y0 + j), icpm), icpm == 4);
          }
      } else if (cat() != CHROMA_MONO) {
        static const uint8_t map422[35] = {0, 1, 2, 2, 2, 2, 3, 5, 7, 8, 10, 12, 13, 15, 17, 18, 19, 20, 21, 22, 23, 23, 24, 24, 25, 25, 26, 27, 27, 28, 28, 29, 29, 30, 31};   /* H.265 Table 8-3 */
        const int icpm = write_icpm();
        int modeC = chroma_mode(img.get_IntraPredMode(x0, y0), icpm);
        if (cat() == CHROMA_422) modeC = map422[modeC];
        img.set_IntraPredModeC(x0, y0, log2, (enum IntraPredMode)modeC, icpm == 4);
      }
    } else {
      const int q = nCbS / 4, h = nCbS / 2;
      switch (part) {
        case PART_2Nx2N: merge_2Nx2N = prediction_unit(nCbS, nCbS, ctDepth); break;
        case PART_2NxN: prediction_unit(nCbS, h, ctDepth); prediction_unit(nCbS, h, ctDepth); break;
        case PART_Nx2N: prediction_unit(h, nCbS, ctDepth); prediction_unit(h, nCbS, ctDepth); break;
        case PART_2NxnU: prediction_unit(nCbS, q, ctDepth); prediction_unit(nCbS, nCbS - q, ctDepth); break;
        case PART_2NxnD: prediction_unit(nCbS, nCbS - q, ctDepth); prediction_unit(nCbS, q, ctDepth); break;
        case PART_nLx2N: prediction_unit(q, nCbS, ctDepth); prediction_unit(nCbS - q, nCbS, ctDepth); break;
        case PART_NxN: for (int k = 0; k < 4; k++) prediction_unit(h, h, ctDepth); break;
        default: prediction_unit(nCbS - q, nCbS, ctDepth); prediction_unit(q, nCbS, ctDepth); break;
      }
    }
    bool root_cbf = true;
    if (!intra && !(part == PART_2Nx2N && merge_2Nx2N)) { root_cbf = pct(70); bit(CONTEXT_MODEL_RQT_ROOT_CBF, root_cbf); }
    if (root_cbf) {
      const int maxDepth = intra ? S.max_transform_hierarchy_depth_intra + intraSplit : S.max_transform_hierarchy_depth_inter;
      const int c0 = cat() == CHROMA_MONO ? 0 : 1;
      transform_tree(x0, y0, x0, y0, log2, 0, 0, maxDepth, intraSplit, intra, part, c0, c0, intra ? 60 : 45);
    }
  }

  /* ---------------- slice segment data: CTBs [ts0, ts1) in tile-scan order; a tile boundary starts a new CABAC substream
     (entry points in the slice header).  `keep_ctx`: a dependent slice segment continues with the context models the segment
     before it ended with (slice.cc:4917-4935). ---------------- */
  void slice_data(CABAC_encoder_bitstream& enc, uint32_t seed, int ts0, int ts1, bool keep_ctx, const context_model_table& start_ctx, std::vector<int>& substream_end)
  {
    cabac = &enc;
    s = seed ? seed : 1;
    const seq_parameter_set& S = *sps;
    const int W = S.PicWidthInCtbsY;
    enc.set_context_models(&ctx);
    /* (a segment that begins a tile starts from fresh models even when it is a dependent one, slice.cc:4925-4929) */
    if (keep_ctx && !(ts0 > 0 && pps->scan->TileId[ts0] != pps->scan->TileId[ts0 - 1])) ctx = start_ctx.copy(); else ctx.init(shdr->initType, shdr->SliceQPY);
    enc.init_CABAC();
    const int start = enc.size();
    substream_end.clear();
    qg_coded = cqo_coded = true;
    const bool wpp = pps->entropy_coding_sync_enabled_flag;
    context_model_table row_ctx;                                               /* WPP: the models after the second CTB of the row above */
    for (int ts = ts0; ts < ts1; ts++) {
      const int rs = pps->scan->CtbAddrTStoRS[ts], xCtb = rs % W, yCtb = rs / W;
      img.set_SliceAddrRS(xCtb, yCtb, shdr->SliceAddrRS);
      img.set_SliceHeaderIndex(xCtb << S.Log2CtbSizeY, yCtb << S.Log2CtbSizeY, slice_index);
      if (shdr->slice_sao_luma_flag || shdr->slice_sao_chroma_flag) write_sao(xCtb, yCtb, rs);
      const int target = 3 + below(S.Log2CtbSizeY - 2);
      coding_quadtree(xCtb << S.Log2CtbSizeY, yCtb << S.Log2CtbSizeY, S.Log2CtbSizeY, 0, target);
      if (wpp && xCtb == 1) row_ctx = ctx.copy();                               /* storage process (9.3.2.2 end; slice.cc:4806-4822) */
      const bool last = ts == ts1 - 1;
      enc.write_CABAC_term_bit(last);                                          /* end_of_slice_segment_flag */
      const bool next_row = wpp && !last && pps->scan->CtbAddrTStoRS[ts + 1] / W != yCtb;
      if (!last && (next_row || pps->scan->TileId[ts + 1] != pps->scan->TileId[ts])) {
        enc.write_CABAC_term_bit(1);                                           /* end_of_subset_one_bit */
        enc.flush_CABAC();
        enc.add_trailing_bits();                                               /* byte_alignment() */
        enc.flush_VLC();
        substream_end.push_back(enc.size() - start);
        /* a new tile starts from fresh models; a new CTB row under WPP from the stored ones when the picture is more than one
           CTB wide (slice.cc:4745-4775) */
        if (next_row && W > 1) ctx = row_ctx.copy(); else ctx.init(shdr->initType, shdr->SliceQPY);
        enc.init_CABAC();
      }
    }
    enc.flush_CABAC();
    enc.add_trailing_bits();                                                   /* rbsp_slice_segment_trailing_bits */
    enc.flush_VLC();
  }

  /* ---------------- 7.3.6.3 pred_weight_table (read_pred_weight_table, slice.cc:150-240) ---------------- */
  void write_pred_weight_table(CABAC_encoder& out, const slice_segment_header* sh)
  {
    const int ldenom = below(8);
    const int cdenom = cat() ? below(8) : ldenom;
    out.write_uvlc(ldenom);
    if (cat()) out.write_svlc(cdenom - ldenom);
    for (int l = 0; l <= (sh->slice_type == SLICE_TYPE_B ? 1 : 0); l++) {
      const int n = nref[l];
      int lf[16], cf[16];
      for (int i = 0; i < n; i++) { lf[i] = pct(75); out.write_bit(lf[i]); }
      if (cat()) for (int i = 0; i < n; i++) { cf[i] = pct(75); out.write_bit(cf[i]); }
      for (int i = 0; i < n; i++) {
        if (lf[i]) { out.write_svlc(range(-((1 << ldenom) / 2 + 4), 24)); out.write_svlc(range(-20, 20)); }   /* delta_luma_weight, luma_offset */
        if (cat() && cf[i]) for (int j = 0; j < 2; j++) { out.write_svlc(range(-((1 << cdenom) / 2 + 4), 24)); out.write_svlc(range(-40, 40)); }   /* delta_chroma_weight, delta_chroma_offset */
      }
    }
  }

  /* ---------------- 7.3.6.1 slice segment header (slice_segment_header::read, slice.cc:388-900).  RPS by SPS index, no
     long-term pictures, no list modification, no temporal MVP (all off in the parameter sets below). ---------------- */
  void write_slice_header(CABAC_encoder& out, const slice_segment_header* sh, int nal_type)
  {
    const seq_parameter_set& S = *sps;
    const pic_parameter_set& P = *pps;
    out.write_bit(sh->first_slice_segment_in_pic_flag);
    if (isRapPic((uint8_t)nal_type)) out.write_bit(0);                          /* no_output_of_prior_pics_flag */
    out.write_uvlc(0);                                                          /* slice_pic_parameter_set_id */
    if (!sh->first_slice_segment_in_pic_flag) {
      if (P.dependent_slice_segments_enabled_flag) out.write_bit(sh->dependent_slice_segment_flag);
      out.write_bits(sh->slice_segment_address, ceil_log2(S.PicSizeInCtbsY));
    }
    if (!sh->dependent_slice_segment_flag) {
      out.write_uvlc(sh->slice_type);
      if (P.output_flag_present_flag) out.write_bit(sh->pic_output_flag);
      if (nal_type != NAL_UNIT_IDR_W_RADL && nal_type != NAL_UNIT_IDR_N_LP) {
        out.write_bits(sh->slice_pic_order_cnt_lsb, S.log2_max_pic_order_cnt_lsb);
        if (has(F_RA)) {
          /* st_ref_pic_set(num_short_term_ref_pic_sets) in the slice header (read_short_term_ref_pic_set, refpic.cc:85-260) */
          out.write_bit(0);                                                     /* short_term_ref_pic_set_sps_flag */
          if (S.num_short_term_ref_pic_sets() != 0) out.write_bit(0);           /* inter_ref_pic_set_prediction_flag */
          out.write_uvlc((int)rps_neg.size()); out.write_uvlc((int)rps_pos.size());
          for (size_t i = 0; i < rps_neg.size(); i++) { out.write_uvlc(rps_neg[i] - (i ? rps_neg[i - 1] : 0) - 1); out.write_bit(rps_neg_used[i]); }
          for (size_t i = 0; i < rps_pos.size(); i++) { out.write_uvlc(rps_pos[i] - (i ? rps_pos[i - 1] : 0) - 1); out.write_bit(rps_pos_used[i]); }
        } else {
          out.write_bit(1);                                                     /* short_term_ref_pic_set_sps_flag */
          const int nb = ceil_log2(S.num_short_term_ref_pic_sets());
          if (nb > 0) out.write_bits(sh->short_term_ref_pic_set_idx, nb);
        }
        if (S.long_term_ref_pics_present_flag) {                                /* slice.cc:517-598; no candidates in the SPS */
          out.write_uvlc((int)lt_lsb.size());
          for (size_t i = 0; i < lt_lsb.size(); i++) { out.write_bits(lt_lsb[i], S.log2_max_pic_order_cnt_lsb); out.write_bit(lt_used[i]); out.write_bit(0); }   /* delta_poc_msb_present_flag */
        }
        if (S.sps_temporal_mvp_enabled_flag) out.write_bit(sh->slice_temporal_mvp_enabled_flag);
      }
      if (S.sample_adaptive_offset_enabled_flag) {
        out.write_bit(sh->slice_sao_luma_flag);
        if (cat() != CHROMA_MONO) out.write_bit(sh->slice_sao_chroma_flag);
      }
      if (sh->slice_type != SLICE_TYPE_I) {
        out.write_bit(1);                                                       /* num_ref_idx_active_override_flag */
        out.write_uvlc(nref[0] - 1);
        if (sh->slice_type == SLICE_TYPE_B) out.write_uvlc(nref[1] - 1);
        if (P.lists_modification_present_flag && list_bits_total > 1) {         /* ref_pic_lists_modification (slice.cc:668-690) */
          const int nb = ceil_log2(list_bits_total);
          for (int l = 0; l <= (sh->slice_type == SLICE_TYPE_B ? 1 : 0); l++) {
            const int mod = pct(75);
            out.write_bit(mod);
            if (mod) for (int i = 0; i < nref[l]; i++) out.write_bits(below(list_total), nb);   /* list_entry_lX: any picture, also twice */
          }
        }
        if (sh->slice_type == SLICE_TYPE_B) out.write_bit(0);                   /* mvd_l1_zero_flag */
        if (P.cabac_init_present_flag) out.write_bit(sh->cabac_init_flag);
        if (sh->slice_temporal_mvp_enabled_flag) {                              /* slice.cc:705-728 */
          if (sh->slice_type == SLICE_TYPE_B) out.write_bit(col_from_l0);
          if (nref[col_from_l0 ? 0 : 1] > 1) out.write_uvlc(col_ref_idx);
        }
        if ((P.weighted_pred_flag && sh->slice_type == SLICE_TYPE_P) || (P.weighted_bipred_flag && sh->slice_type == SLICE_TYPE_B)) write_pred_weight_table(out, sh);
        out.write_uvlc(sh->five_minus_max_num_merge_cand);
      }
      out.write_svlc(sh->slice_qp_delta);
      if (P.pps_slice_chroma_qp_offsets_present_flag) { out.write_svlc(sh->slice_cb_qp_offset); out.write_svlc(sh->slice_cr_qp_offset); }
      if (P.range_extension.chroma_qp_offset_list_enabled_flag) out.write_bit(sh->cu_chroma_qp_offset_enabled_flag);
      if (P.deblocking_filter_override_enabled_flag) out.write_bit(sh->deblocking_filter_override_flag);
      if (sh->deblocking_filter_override_flag) {
        out.write_bit(sh->slice_deblocking_filter_disabled_flag);
        if (!sh->slice_deblocking_filter_disabled_flag) { out.write_svlc(sh->slice_beta_offset / 2); out.write_svlc(sh->slice_tc_offset / 2); }
      }
      if (P.pps_loop_filter_across_slices_enabled_flag && (sh->slice_sao_luma_flag || sh->slice_sao_chroma_flag || !sh->slice_deblocking_filter_disabled_flag))
        out.write_bit(sh->slice_loop_filter_across_slices_enabled_flag);
    }
    if (P.tiles_enabled_flag || P.entropy_coding_sync_enabled_flag) {
      out.write_uvlc(sh->num_entry_point_offsets);
      if (sh->num_entry_point_offsets > 0) {
        out.write_uvlc(sh->offset_len - 1);
        for (int i = 0; i < sh->num_entry_point_offsets; i++) out.write_bits(sh->entry_point_offset[i] - (i ? sh->entry_point_offset[i - 1] : 0) - 1, sh->offset_len);
      }
    }
  }

  /* ---------------- 7.3.4 scaling_list_data (read_scaling_list, sps.cc:939-1081): per size and matrix either a reference
     (0 = the default list, d = the d-th matrix before this one) or 16 / 64 coefficients as differences (+ the DC of 16x16 / 32x32) */
  void write_scaling_list_data(CABAC_encoder& out)
  {
    for (int sizeId = 0; sizeId < 4; sizeId++)
      for (int matrixId = 0; matrixId < 6; matrixId += sizeId == 3 ? 3 : 1) {
        const int explicit_ = pct(60);
        out.write_bit(explicit_);                                               /* scaling_list_pred_mode_flag */
        if (!explicit_) { out.write_uvlc(below((sizeId == 3 ? matrixId / 3 : matrixId) + 1)); continue; }
        const int n = sizeId == 0 ? 16 : 64;
        int next = 8;
        if (sizeId > 1) { const int dc = range(4, 40); out.write_svlc(dc - 8); next = dc; }
        for (int i = 0; i < n; i++) {
          const int v = std::min(255, std::max(1, 8 + i / 2 + range(-4, 12)));  /* rising with frequency, like real lists */
          int d = v - next;
          if (d > 127) d -= 256; else if (d < -128) d += 256;
          out.write_svlc(d);
          next = v;
        }
      }
  }

  /* ---------------- 7.3.2.3 picture parameter set incl. pps_range_extension (pic_parameter_set::read, pps.cc:280-560,
     pps_range_extension::read pps.cc:50-143) ---------------- */
  void write_pps(CABAC_encoder& out)
  {
    const pic_parameter_set& P = *pps;
    out.write_uvlc(0); out.write_uvlc(0);                                       /* pps / sps id */
    out.write_bit(P.dependent_slice_segments_enabled_flag);
    out.write_bit(P.output_flag_present_flag);
    out.write_bits(0, 3);                                                       /* num_extra_slice_header_bits */
    out.write_bit(P.sign_data_hiding_flag);
    out.write_bit(P.cabac_init_present_flag);
    out.write_uvlc(P.num_ref_idx_l0_default_active - 1); out.write_uvlc(P.num_ref_idx_l1_default_active - 1);
    out.write_svlc(P.pic_init_qp - 26);
    out.write_bit(P.constrained_intra_pred_flag);
    out.write_bit(P.transform_skip_enabled_flag);
    out.write_bit(P.cu_qp_delta_enabled_flag);
    if (P.cu_qp_delta_enabled_flag) out.write_uvlc(P.diff_cu_qp_delta_depth);
    out.write_svlc(P.pic_cb_qp_offset); out.write_svlc(P.pic_cr_qp_offset);
    out.write_bit(P.pps_slice_chroma_qp_offsets_present_flag);
    out.write_bit(P.weighted_pred_flag); out.write_bit(P.weighted_bipred_flag);
    out.write_bit(P.transquant_bypass_enable_flag);
    out.write_bit(P.tiles_enabled_flag);
    out.write_bit(P.entropy_coding_sync_enabled_flag);
    if (P.tiles_enabled_flag) {
      out.write_uvlc(P.num_tile_columns - 1); out.write_uvlc(P.num_tile_rows - 1);
      out.write_bit(P.uniform_spacing_flag);
      if (!P.uniform_spacing_flag) {                                            /* the last column / row is what is left (pps.cc:392-421) */
        for (int i = 0; i < P.num_tile_columns - 1; i++) out.write_uvlc(P.colWidth[i] - 1);
        for (int i = 0; i < P.num_tile_rows - 1; i++) out.write_uvlc(P.rowHeight[i] - 1);
      }
      out.write_bit(P.loop_filter_across_tiles_enabled_flag);
    }
    out.write_bit(P.pps_loop_filter_across_slices_enabled_flag);
    out.write_bit(P.deblocking_filter_control_present_flag);
    if (P.deblocking_filter_control_present_flag) {
      out.write_bit(P.deblocking_filter_override_enabled_flag);
      out.write_bit(P.pic_disable_deblocking_filter_flag);
      if (!P.pic_disable_deblocking_filter_flag) { out.write_svlc(P.beta_offset / 2); out.write_svlc(P.tc_offset / 2); }
    }
    out.write_bit(P.pic_scaling_list_data_present_flag);
    if (P.pic_scaling_list_data_present_flag) write_scaling_list_data(out);
    out.write_bit(P.lists_modification_present_flag);
    out.write_uvlc(P.log2_parallel_merge_level - 2);
    out.write_bit(0);                                                           /* slice_segment_header_extension_present_flag */
    out.write_bit(P.pps_range_extension_flag);                                  /* pps_extension_present_flag */
    if (P.pps_range_extension_flag) {
      out.write_bit(1); out.write_bit(0); out.write_bits(0, 6);                 /* range / multilayer / 6 more extension flags */
      const pps_range_extension& R = P.range_extension;
      if (P.transform_skip_enabled_flag) out.write_uvlc(R.log2_max_transform_skip_block_size - 2);
      out.write_bit(R.cross_component_prediction_enabled_flag);
      out.write_bit(R.chroma_qp_offset_list_enabled_flag);
      if (R.chroma_qp_offset_list_enabled_flag) {
        out.write_uvlc(R.diff_cu_chroma_qp_offset_depth);
        out.write_uvlc(R.chroma_qp_offset_list_len - 1);
        for (int i = 0; i < R.chroma_qp_offset_list_len; i++) { out.write_svlc(R.cb_qp_offset_list[i]); out.write_svlc(R.cr_qp_offset_list[i]); }
      }
      out.write_uvlc(0); out.write_uvlc(0);                                     /* log2_sao_offset_scale_luma / _chroma */
    }
  }
};

int run(const Cfg& cfg, const char* out_name)
{
  de265_init();                                /* scan-order tables (scan.cc) used by the residual writer */
  std::unique_ptr<Gen> G(new Gen);
  Gen& g = *G;
  g.cfg = cfg;
  const bool plain = cfg.features == 0 && cfg.chroma == 1 && cfg.slices == 1 && cfg.geom == 0;  /* then every header comes from the reference's own writers */
  const bool rext = g.has(F_REXT);
  g.vps = std::make_shared<video_parameter_set>();
  g.sps = std::make_shared<seq_parameter_set>();
  g.pps = std::make_shared<pic_parameter_set>();
  g.vps->set_defaults(Profile_Main, 6, 2);
  seq_parameter_set& S = *g.sps;
  S.set_defaults();
  const int ctbLog2 = 6 - (cfg.geom & G_CTB);
  const int maxTbLog2 = std::min((cfg.geom & G_TB16) ? 4 : 5, ctbLog2);
  S.set_CB_log2size_range((cfg.geom & G_MINCB16) ? 4 : 3, ctbLog2);
  S.set_TB_log2size_range(2, maxTbLog2);
  S.set_resolution(cfg.W, cfg.H);
  if (cfg.geom & G_CONFWIN) {                                                   /* offsets count chroma samples (sps.cc:228-262) */
    S.conformance_window_flag = 1;
    const int sw = cfg.chroma == 1 || cfg.chroma == 2 ? 2 : 1, sh = cfg.chroma == 1 ? 2 : 1;
    S.conf_win_left_offset = 4 / sw; S.conf_win_right_offset = 8 / sw; S.conf_win_top_offset = 2 / sh; S.conf_win_bottom_offset = 6 / sh;
  }
  S.chroma_format_idc = cfg.chroma;
  S.bit_depth_luma = S.bit_depth_chroma = cfg.bd;
  S.log2_max_pic_order_cnt_lsb = 8;
  S.sps_max_dec_pic_buffering[0] = 4; S.sps_max_num_reorder_pics[0] = 0; S.sps_max_latency_increase_plus1[0] = 0;
  if (g.has(F_RA)) { S.sps_max_dec_pic_buffering[0] = 8; S.sps_max_num_reorder_pics[0] = 3; }   /* up to 6 reference pictures + the current one; 8 4 2 precede picture 1 */
  S.max_transform_hierarchy_depth_inter = 2;
  S.max_transform_hierarchy_depth_intra = 2;
  S.amp_enabled_flag = 1;
  S.sample_adaptive_offset_enabled_flag = cfg.sao ? 1 : 0;
  S.pcm_enabled_flag = 0;
  if (g.has(F_PCM)) {
    S.pcm_enabled_flag = 1;
    S.pcm_sample_bit_depth_luma = cfg.bd - 1; S.pcm_sample_bit_depth_chroma = cfg.bd - 2;     /* samples are shifted up to the bit depth (slice.cc:4243-4253) */
    S.log2_min_pcm_luma_coding_block_size = (cfg.geom & G_MINCB16) ? 4 : 3;      /* >= the smallest coding block, <= min(CTB, 32) */
    S.log2_diff_max_min_pcm_luma_coding_block_size = std::min(5, ctbLog2) - S.log2_min_pcm_luma_coding_block_size;
    S.pcm_loop_filter_disable_flag = (cfg.seed >> 1) & 1;
  }
  if (g.has(F_SCALING) || g.has(F_SCALING_PPS)) { S.scaling_list_enable_flag = 1; S.sps_scaling_list_data_present_flag = 0; }   /* the SPS carries the default lists */
  S.long_term_ref_pics_present_flag = g.has(F_LT) ? 1 : 0;
  S.num_long_term_ref_pics_sps = 0;
  S.sps_temporal_mvp_enabled_flag = g.has(F_TMVP) ? 1 : 0;
  S.strong_intra_smoothing_enable_flag = (cfg.geom & G_TB16) ? 0 : 1;
  if (rext) {
    S.sps_extension_present_flag = 1; S.sps_range_extension_flag = 1;
    S.range_extension.transform_skip_rotation_enabled_flag = 1;
    S.range_extension.implicit_rdpcm_enabled_flag = 1;
    S.range_extension.explicit_rdpcm_enabled_flag = 1;
    S.range_extension.intra_smoothing_disabled_flag = (cfg.seed >> 2) & 1;
  }
  S.ref_pic_sets.resize(2);
  for (int k = 0; k < 2; k++) {
    ref_pic_set& r = S.ref_pic_sets[k];
    r.reset();
    r.NumNegativePics = k + 1; r.NumPositivePics = 0;
    for (int i = 0; i <= k; i++) { r.DeltaPocS0[i] = -(i + 1); r.UsedByCurrPicS0[i] = 1; }
    r.compute_derived_values();
  }
  if (S.compute_derived_values() != DE265_OK) { fprintf(stderr, "streamgen: bad SPS\n"); return 2; }
  S.sps_read = true;
  pic_parameter_set& P = *g.pps;
  P.set_defaults();
  P.sps = g.sps;
  P.pic_init_qp = 30;
  P.num_ref_idx_l0_default_active = 1; P.num_ref_idx_l1_default_active = 1;
  P.tiles_enabled_flag = (cfg.tc > 1 || cfg.tr > 1);
  P.num_tile_columns = cfg.tc; P.num_tile_rows = cfg.tr;
  P.uniform_spacing_flag = 1;
  if ((cfg.geom & G_TILES) && P.tiles_enabled_flag) {
    /* random cuts: every column / row at least one CTB (set_derived_values takes colWidth / rowHeight as they are then, pps.cc:696-731) */
    P.uniform_spacing_flag = 0;
    uint32_t tr = cfg.seed * 747796405u + 2891336453u;
    auto r = [&]() { tr ^= tr << 13; tr ^= tr >> 17; tr ^= tr << 5; return tr; };
    auto cutup = [&](int total, int parts, uint16_t* size) {
      for (int i = 0; i < parts; i++) size[i] = 1;
      for (int left = total - parts; left > 0; left--) size[r() % parts]++;
    };
    cutup(S.PicWidthInCtbsY, cfg.tc, P.colWidth);
    cutup(S.PicHeightInCtbsY, cfg.tr, P.rowHeight);
  }
  P.loop_filter_across_tiles_enabled_flag = (cfg.geom & G_NOTILEFILTER) ? 0 : 1;
  if (cfg.geom & G_PARMERGE) P.log2_parallel_merge_level = std::min(4, ctbLog2);
  P.pps_loop_filter_across_slices_enabled_flag = 1;
  P.deblocking_filter_control_present_flag = 0;
  P.pic_cb_qp_offset = 1; P.pic_cr_qp_offset = -1;
  P.sign_data_hiding_flag = g.has(F_SDH);
  P.entropy_coding_sync_enabled_flag = g.has(F_WPP);
  if (!plain) {
    P.constrained_intra_pred_flag = g.has(F_CIP);
    P.transform_skip_enabled_flag = g.has(F_TSKIP);
    P.transquant_bypass_enable_flag = g.has(F_BYPASS);
    P.cu_qp_delta_enabled_flag = g.has(F_QPDELTA); P.diff_cu_qp_delta_depth = g.has(F_QPDELTA) ? std::min(1 + (int)(cfg.seed & 1), (int)S.log2_diff_max_min_luma_coding_block_size) : 0;
    P.pps_slice_chroma_qp_offsets_present_flag = g.has(F_QPDELTA) && cfg.chroma != 0;
    P.weighted_pred_flag = P.weighted_bipred_flag = g.has(F_WP);
    P.dependent_slice_segments_enabled_flag = g.has(F_DEPSLICE);
    P.cabac_init_present_flag = g.has(F_MIXSLICE);
    P.lists_modification_present_flag = g.has(F_LISTMOD);
    P.output_flag_present_flag = g.has(F_NOOUTPUT);
    if (cfg.slices > 1) { P.deblocking_filter_control_present_flag = 1; P.deblocking_filter_override_enabled_flag = 1; P.pic_disable_deblocking_filter_flag = 0; P.beta_offset = 2; P.tc_offset = -2; }
    P.pic_scaling_list_data_present_flag = g.has(F_SCALING_PPS);
    if (rext) {
      P.pps_extension_flag = 1; P.pps_range_extension_flag = 1;
      P.range_extension.log2_max_transform_skip_block_size = g.has(F_TSKIP) ? maxTbLog2 : 2;
      P.range_extension.cross_component_prediction_enabled_flag = cfg.chroma == 3;
      if (cfg.chroma != 0) {
        P.range_extension.chroma_qp_offset_list_enabled_flag = 1;
        P.range_extension.diff_cu_chroma_qp_offset_depth = std::min(1, (int)S.log2_diff_max_min_luma_coding_block_size);
        P.range_extension.chroma_qp_offset_list_len = 2;
        P.range_extension.cb_qp_offset_list[0] = 3; P.range_extension.cr_qp_offset_list[0] = -4;
        P.range_extension.cb_qp_offset_list[1] = -6; P.range_extension.cr_qp_offset_list[1] = 5;
      }
    }
  }
  P.set_derived_values(g.sps.get());
  P.pps_read = true;

  if (g.img.alloc_image(cfg.W, cfg.H, (de265_chroma)cfg.chroma, g.sps, true, &g.dctx, 0, nullptr, false) != DE265_OK) return 2;
  g.img.set_headers(g.vps, g.sps, g.pps);
  g.ectx.img = &g.img;

  FILE* f = fopen(out_name, "wb");
  if (!f) return 2;
  CABAC_encoder_bitstream out;
  nal_header nal;
  out.write_startcode(); nal.set(NAL_UNIT_VPS_NUT); nal.write(out); g.vps->write(&g.dctx, out); out.add_trailing_bits(); out.flush_VLC();
  out.write_startcode(); nal.set(NAL_UNIT_SPS_NUT); nal.write(out); g.sps->write(&g.dctx, out);
  if (rext) {
    /* sps_range_extension (sps_range_extension::read, sps.cc:1357-1371): the reference's writer stops at sps_extension_present_flag */
    const sps_range_extension& R = S.range_extension;
    out.write_bit(1); out.write_bit(0); out.write_bits(0, 6);                   /* range / multilayer / 6 more extension flags */
    out.write_bit(R.transform_skip_rotation_enabled_flag); out.write_bit(R.transform_skip_context_enabled_flag);
    out.write_bit(R.implicit_rdpcm_enabled_flag); out.write_bit(R.explicit_rdpcm_enabled_flag);
    out.write_bit(R.extended_precision_processing_flag); out.write_bit(R.intra_smoothing_disabled_flag);
    out.write_bit(R.high_precision_offsets_enabled_flag); out.write_bit(R.persistent_rice_adaptation_enabled_flag);
    out.write_bit(R.cabac_bypass_alignment_enabled_flag);
  }
  out.add_trailing_bits(); out.flush_VLC();
  out.write_startcode(); nal.set(NAL_UNIT_PPS_NUT); nal.write(out);
  if (plain) g.pps->write(&g.dctx, out, g.sps.get()); else { g.s = cfg.seed * 7919u + 13u; g.write_pps(out); }
  out.add_trailing_bits(); out.flush_VLC();
  fwrite(out.data(), 1, out.size(), f);

  const int nCtb = S.PicSizeInCtbsY;
  /* coding order.  Plain: POC = picture index, I then P / B pictures on the one or two pictures before (RPS from the SPS).
     F_RA: groups of 8 in the order 8 4 2 1 3 6 5 7; a picture keeps, as short-term references, the two anchors around it, the
     even pictures of its group decoded so far and (an anchor only) the anchor before the last; odd pictures are never referenced.
     F_LT: from the second group's inner pictures on the IDR picture is listed as a long-term reference instead of being dropped. */
  struct Plan { int poc, nal, type, nrefs; std::vector<int> neg, neg_used, pos, pos_used, lt_lsb, lt_used; };
  std::vector<Plan> plans;
  if (!g.has(F_RA)) {
    for (int fr = 0; fr < cfg.frames; fr++) {
      Plan pl;
      pl.poc = fr; pl.nal = fr == 0 ? NAL_UNIT_IDR_W_RADL : NAL_UNIT_TRAIL_R;
      pl.type = fr == 0 ? SLICE_TYPE_I : ((cfg.b_frames && (fr & 1) == 0) ? SLICE_TYPE_B : SLICE_TYPE_P);
      pl.nrefs = fr >= 2 ? 2 : fr;
      plans.push_back(pl);
    }
  } else {
    static const int gop[8] = {8, 4, 2, 1, 3, 6, 5, 7};
    std::vector<int> order(1, 0);
    for (int base = 0; (int)order.size() < cfg.frames; base += 8)
      for (int k = 0; k < 8 && (int)order.size() < cfg.frames; k++) order.push_back(base + gop[k]);
    std::vector<int> alive;                                                     /* POCs of the short-term reference pictures in the DPB */
    bool lt0 = false;                                                           /* POC 0 has become a long-term picture */
    uint32_t pr = cfg.seed * 69069u + 12345u;
    auto r = [&]() { pr ^= pr << 13; pr ^= pr >> 17; pr ^= pr << 5; return pr; };
    for (int p : order) {
      Plan pl;
      pl.poc = p;
      if (p == 0) { pl.nal = NAL_UNIT_IDR_W_RADL; pl.type = SLICE_TYPE_I; pl.nrefs = 0; alive.assign(1, 0); plans.push_back(pl); continue; }
      pl.nal = (p & 1) ? NAL_UNIT_TRAIL_N : NAL_UNIT_TRAIL_R;
      pl.type = SLICE_TYPE_B;
      const int a1 = (p + 7) / 8 * 8, a0 = a1 - 8;
      std::vector<int> keep;
      bool had0 = false;
      for (int q : alive) {
        const bool k = q == a1 || q == a0 || (q > a0 && q < a1) || (p == a1 && q == a0 - 8);
        if (k) keep.push_back(q); else if (q == 0) had0 = true;
      }
      if (g.has(F_LT) && (had0 || lt0)) { lt0 = true; pl.lt_lsb.push_back(0); pl.lt_used.push_back(r() % 10 < 7); }
      std::sort(keep.begin(), keep.end());
      for (int i = (int)keep.size() - 1; i >= 0; i--) if (keep[i] < p) { pl.neg.push_back(p - keep[i]); pl.neg_used.push_back(1); }
      for (size_t i = 0; i < keep.size(); i++) if (keep[i] > p) { pl.pos.push_back(keep[i] - p); pl.pos_used.push_back(1); }
      /* some pictures are kept for later pictures only (RefPicSetStFoll): the farthest one before, now and then */
      if (pl.neg.size() + pl.pos.size() >= 3 && pl.neg.size() >= 2 && r() % 3 == 0) pl.neg_used.back() = 0;
      pl.nrefs = 4;                                                              /* lists longer than the set repeat its pictures (decctx.cc:1580-1640) */
      alive = keep;
      if (!(p & 1)) alive.push_back(p);
      plans.push_back(pl);
    }
  }
  for (size_t pi = 0; pi < plans.size(); pi++) {
    const Plan& plan = plans[pi];
    const int fr = (int)pi;                                                     /* coding index: seeds */
    const int type = plan.type, nal_type = plan.nal, nrefs = plan.nrefs;
    g.slice_type = type; g.nref[0] = g.nref[1] = nrefs;
    g.rps_neg = plan.neg; g.rps_neg_used = plan.neg_used; g.rps_pos = plan.pos; g.rps_pos_used = plan.pos_used;
    g.lt_lsb = plan.lt_lsb; g.lt_used = plan.lt_used;
    /* slice segments: [cut[k], cut[k+1]) in tile-scan order; with tiles a segment holds whole tiles (7.4.7.1) */
    std::vector<int> cut;
    cut.push_back(0);
    if (cfg.slices > 1) {
      uint32_t sr = cfg.seed * 2246822519u + 3266489917u * (uint32_t)(fr + 1);
      auto r = [&]() { sr ^= sr << 13; sr ^= sr >> 17; sr ^= sr << 5; return sr; };
      std::vector<int> cand;
      for (int ts = 1; ts < nCtb; ts++) if (!P.tiles_enabled_flag || P.scan->TileId[ts] != P.scan->TileId[ts - 1]) cand.push_back(ts);
      for (int k = 1; k < cfg.slices && !cand.empty(); k++) { const size_t i = r() % cand.size(); cut.push_back(cand[i]); cand.erase(cand.begin() + i); }
      std::sort(cut.begin(), cut.end());
    }
    cut.push_back(nCtb);
    g.img.clear_metadata();
    for (slice_segment_header* h : g.img.slices) delete h;
    g.img.slices.clear();
    context_model_table end_ctx;                                               /* context models at the end of the previous segment */
    int stype = type, snref = nrefs;                                           /* of the slice the current segment belongs to */
    {
      /* pictures a list may name = NumPocTotalCurr: the short-term pictures "used by curr" and the used long-term ones (slice.cc:574-577, 665) */
      int st = 0, lt_used = 0;
      if (g.has(F_RA)) { for (int u : plan.neg_used) st += u; for (int u : plan.pos_used) st += u; for (int u : plan.lt_used) lt_used += u; }
      else st = nrefs >= 2 ? 2 : nrefs;
      g.list_total = g.list_bits_total = st + lt_used;
    }
    uint32_t indep_addr = 0;
    slice_segment_header* indep = nullptr;
    for (size_t k = 0; k + 1 < cut.size(); k++) {
      slice_segment_header* sh = new slice_segment_header;                    /* owned by the image (img.slices) */
      g.shdr = sh;
      g.slice_index = (int)g.img.slices.size();
      g.img.slices.push_back(sh);
      const bool dependent = g.has(F_DEPSLICE) && k > 0 && (k & 1);
      uint32_t hr = cfg.seed * 40503u + 2654435761u * (uint32_t)(fr * 64 + (int)k + 1);
      auto r = [&]() { hr ^= hr << 13; hr ^= hr >> 17; hr ^= hr << 5; return hr; };
      if (dependent) *sh = *indep;                                              /* a dependent segment inherits the header of its slice (slice.cc:424-444) */
      sh->first_slice_segment_in_pic_flag = k == 0;
      sh->dependent_slice_segment_flag = dependent;
      sh->slice_segment_address = P.scan->CtbAddrTStoRS[cut[k]];
      sh->slice_pic_parameter_set_id = 0;
      sh->pps = g.pps;
      if (!dependent) {
        indep_addr = sh->slice_segment_address;
        /* this slice's own type and number of active references (F_MIXSLICE): I / P slices inside P / B pictures */
        stype = type; snref = nrefs;
        if (g.has(F_MIXSLICE) && type != SLICE_TYPE_I) {
          const uint32_t q = r() % 10;
          if (q < 2) stype = SLICE_TYPE_I; else if (q < 5 && type == SLICE_TYPE_B) stype = SLICE_TYPE_P;
          snref = 1 + (int)(r() % (uint32_t)nrefs);
        }
        g.slice_type = stype; g.nref[0] = g.nref[1] = stype == SLICE_TYPE_I ? 0 : snref;
        sh->slice_type = stype;
        sh->pic_output_flag = (g.has(F_NOOUTPUT) && fr > 0 && (plan.poc % 3) == 1) ? 0 : 1;   /* (the same in every slice of the picture) */
        sh->cabac_init_flag = (g.has(F_MIXSLICE) && stype != SLICE_TYPE_I) ? (int)(r() % 2) : 0;
        sh->slice_pic_order_cnt_lsb = plan.poc & 0xFF;
        sh->short_term_ref_pic_set_sps_flag = g.has(F_RA) ? 0 : 1;
        sh->short_term_ref_pic_set_idx = nrefs >= 2 ? 1 : 0;
        sh->slice_temporal_mvp_enabled_flag = (g.has(F_TMVP) && stype != SLICE_TYPE_I) ? (r() % 8 != 0) : 0;
        g.col_from_l0 = stype == SLICE_TYPE_B ? (int)(r() % 2) : 1;
        g.col_ref_idx = snref > 1 ? (int)(r() % (uint32_t)snref) : 0;
        sh->slice_sao_luma_flag = sh->slice_sao_chroma_flag = cfg.sao ? 1 : 0;
        sh->num_ref_idx_active_override_flag = stype != SLICE_TYPE_I;
        sh->num_ref_idx_l0_active = snref; sh->num_ref_idx_l1_active = snref;
        sh->five_minus_max_num_merge_cand = g.has(F_MIXSLICE) ? (int)(r() % 5) : 0;
        sh->slice_qp_delta = 0;
        sh->slice_loop_filter_across_slices_enabled_flag = 1;
        sh->slice_deblocking_filter_disabled_flag = P.pic_disable_deblocking_filter_flag;
        sh->slice_beta_offset = P.beta_offset; sh->slice_tc_offset = P.tc_offset;
        if (!plain) {
          if (cfg.chroma == 0) sh->slice_sao_chroma_flag = 0;
          if (g.has(F_QPDELTA)) { sh->slice_qp_delta = (int)(r() % 13) - 6; if (cfg.chroma) { sh->slice_cb_qp_offset = (int)(r() % 7) - 3; sh->slice_cr_qp_offset = (int)(r() % 7) - 3; } }
          if (P.range_extension.chroma_qp_offset_list_enabled_flag) sh->cu_chroma_qp_offset_enabled_flag = 1;
          if (cfg.slices > 1) {
            /* every slice its own in-loop filter parameters (what deblock.cc:184-216, 521-523 and sao.cc:122-164, 367 read) */
            if (cfg.sao) { sh->slice_sao_luma_flag = r() % 4 != 0; sh->slice_sao_chroma_flag = cfg.chroma ? r() % 4 != 0 : 0; }
            sh->deblocking_filter_override_flag = r() % 4 != 0;
            if (sh->deblocking_filter_override_flag) {
              sh->slice_deblocking_filter_disabled_flag = r() % 4 == 0;
              if (!sh->slice_deblocking_filter_disabled_flag) { sh->slice_beta_offset = 2 * ((int)(r() % 7) - 3); sh->slice_tc_offset = 2 * ((int)(r() % 7) - 3); }
            }
            sh->slice_loop_filter_across_slices_enabled_flag = r() % 3 != 0;
          }
        }
        sh->SliceAddrRS = sh->slice_segment_address;
        sh->compute_derived_values(g.pps.get());
        sh->MaxNumMergeCand = 5 - sh->five_minus_max_num_merge_cand;
        sh->SliceQPY = P.pic_init_qp + sh->slice_qp_delta;
        sh->initType = stype == SLICE_TYPE_I ? 0 : (stype == SLICE_TYPE_P ? 1 + sh->cabac_init_flag : 2 - sh->cabac_init_flag);   /* slice.cc:1278-1283 */
        indep = sh;
      } else sh->SliceAddrRS = indep_addr;
      const uint32_t seed = cfg.seed * 2654435761u + 977u * (uint32_t)fr + 131071u * (uint32_t)k + 1u;

      /* pass 1: the slice data alone, for the entry points of the tile substreams */
      std::vector<int> ends;
      { CABAC_encoder_bitstream scratch; g.slice_data(scratch, seed, cut[k], cut[k + 1], dependent, end_ctx, ends); }
      sh->num_entry_point_offsets = (int)ends.size();
      sh->entry_point_offset.clear();
      int maxd = 1;
      for (size_t i = 0; i < ends.size(); i++) { const int d = ends[i] - (i ? ends[i - 1] : 0); if (d > maxd) maxd = d; sh->entry_point_offset.push_back((uint32_t)ends[i]); }
      sh->offset_len = 1; while ((1 << sh->offset_len) < maxd) sh->offset_len++;
      /* pass 2: NAL header, slice header, the same slice data */
      CABAC_encoder_bitstream enc;
      enc.write_startcode();
      nal.set(nal_type); nal.write(enc);
      if (plain) {
        /* the header writer takes num_ref_idx_lX_active as the syntax element (minus 1) and leaves the count behind */
        sh->num_ref_idx_l0_active = snref - 1; sh->num_ref_idx_l1_active = snref - 1;
        if (sh->write(&g.dctx, enc, g.sps.get(), g.pps.get(), (uint8_t)nal_type) != DE265_OK) { fprintf(stderr, "streamgen: slice header not writable\n"); return 2; }
      } else {
        g.s = seed ^ 0x9E3779B9u;                                              /* (the weight table draws from the generator) */
        g.write_slice_header(enc, sh, nal_type);
      }
      if (stype == SLICE_TYPE_I) { sh->num_ref_idx_l0_active = sh->num_ref_idx_l1_active = 0; }
      if (stype == SLICE_TYPE_P) { sh->num_ref_idx_l0_active = snref; sh->num_ref_idx_l1_active = 0; }
      if (stype == SLICE_TYPE_B) { sh->num_ref_idx_l0_active = sh->num_ref_idx_l1_active = snref; }
      enc.add_trailing_bits();                                                   /* byte_alignment() of the slice header */
      enc.flush_VLC();
      std::vector<int> ends2;
      g.slice_data(enc, seed, cut[k], cut[k + 1], dependent, end_ctx, ends2);
      if (ends2 != ends) { fprintf(stderr, "streamgen: substream sizes changed between the passes\n"); return 3; }
      end_ctx = g.ctx.copy();
      fwrite(enc.data(), 1, enc.size(), f);
    }
    for (slice_segment_header* h : g.img.slices) delete h;
    g.img.slices.clear();
  }
  fclose(f);
  return 0;
}

} // namespace

int main(int argc, char** argv)
{
  if (argc < 9) { fprintf(stderr, "usage: %s out.h265 W H bit_depth tile_cols tile_rows n_frames seed [intra_pct=5] [b_frames=1] [sao=1] [features=0] [chroma=1] [slices=1] [geom=0]\n", argv[0]); return 2; }
  Cfg c;
  c.W = atoi(argv[2]); c.H = atoi(argv[3]); c.bd = atoi(argv[4]); c.tc = atoi(argv[5]); c.tr = atoi(argv[6]); c.frames = atoi(argv[7]);
  c.seed = (uint32_t)strtoul(argv[8], nullptr, 0);
  c.intra_pct = argc > 9 ? atoi(argv[9]) : 5; c.b_frames = argc > 10 ? atoi(argv[10]) : 1; c.sao = argc > 11 ? atoi(argv[11]) : 1;
  c.features = argc > 12 ? (int)strtol(argv[12], nullptr, 0) : 0; c.chroma = argc > 13 ? atoi(argv[13]) : 1; c.slices = argc > 14 ? atoi(argv[14]) : 1;
  c.geom = argc > 15 ? (int)strtol(argv[15], nullptr, 0) : 0;
  const int min_cb = (c.geom & G_MINCB16) ? 16 : 8;
  if ((c.geom & G_CTB) == 3 || c.W % min_cb || c.H % min_cb || c.geom < 0 || c.geom > 255) { fprintf(stderr, "streamgen: bad geometry\n"); return 2; }
  if (c.tc > (c.W + (64 >> (c.geom & G_CTB)) - 1) / (64 >> (c.geom & G_CTB)) || c.tr > (c.H + (64 >> (c.geom & G_CTB)) - 1) / (64 >> (c.geom & G_CTB))) { fprintf(stderr, "streamgen: more tiles than CTBs\n"); return 2; }
  if (c.W % 8 || c.H % 8 || c.W < 16 || c.H < 16 || c.bd < 8 || c.bd > 12 || c.tc < 1 || c.tr < 1 || c.frames < 1 || c.chroma < 0 || c.chroma > 3 || c.slices < 1 || c.slices > 32) { fprintf(stderr, "streamgen: bad arguments\n"); return 2; }
  if ((c.features & F_PCM) && c.bd - 2 < 1) { fprintf(stderr, "streamgen: bad arguments\n"); return 2; }
  /* WPP: one slice, no tiles (the reference refuses tiles + WPP with threads, decctx.cc:813-817); the reference picture set and the
     collocated picture are per picture: one slice; the residual writer knows the hidden sign only outside the range extensions */
  if (((c.features & F_WPP) && (c.tc > 1 || c.tr > 1 || c.slices > 1)) || ((c.features & (F_RA | F_TMVP)) && c.slices > 1) || ((c.features & F_LT) && !(c.features & F_RA)) ||
      ((c.features & F_SDH) && (c.features & F_REXT)) || ((c.features & F_RA) && c.frames > 200)) { fprintf(stderr, "streamgen: bad feature combination\n"); return 2; }
  return run(c, argv[1]);
}
