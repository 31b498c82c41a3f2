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
 *
 * usage: streamgen out.h265 W H bit_depth tile_cols tile_rows n_frames seed [intra_pct=5] [b_frames=1] [sao=1]
 *                  [features=0] [chroma=1] [slices=1]
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

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

enum { F_WP = 1, F_TSKIP = 2, F_BYPASS = 4, F_QPDELTA = 8, F_PCM = 16, F_SCALING = 32, F_SCALING_PPS = 64, F_REXT = 256, F_CIP = 512, F_DEPSLICE = 1024 };
struct Cfg { int W, H, bd, tc, tr, frames; uint32_t seed; int intra_pct, b_frames, sao, features, chroma, slices; };

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

  uint32_t rnd() { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return s; }
  int below(int n) { return (int)(rnd() % (uint32_t)n); }
  int range(int lo, int hi) { return lo + below(hi - lo + 1); }
  bool pct(int p) { return below(100) < p; }

  void bit(int model, int b) { cabac->write_CABAC_bit(model, b); }
  void bypass(int b) { cabac->write_CABAC_bypass(b); }

  /* ---------------- 7.3.8.3 sample adaptive offset (mirrors read_sao, slice.cc:2735-2871) ---------------- */
  void write_sao(int xCtb, int yCtb, int ctbAddrRS)
  {
    const seq_parameter_set& S = *sps;
    const int W = S.PicWidthInCtbsY;
    sao_info info; memset(&info, 0, sizeof(info));
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
      for (int c = 0; c < 3; c++) {
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
    (void)info;
  }

  /* ---------------- 7.3.8.4 coding quadtree (read_coding_quadtree, slice.cc:4640-4720) ---------------- */
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

  /* ---------------- residual: random sparse block through the reference's own residual_coding writer ---------------- */
  void residual(int x0, int y0, int log2, int cIdx, bool intra)
  {
    const int n = 1 << log2;
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
    encode_residual(&ectx, cabac, &tb, &cb, x0, y0, log2, cIdx);
  }

  /* ---------------- 7.3.8.8 / 7.3.8.10 transform tree + unit (slice.cc:3870-4025, 3584-3860; 4:2:0) ---------------- */
  void transform_tree(int x0, int y0, int xBase, int yBase, int log2, int depth, int blkIdx, int maxDepth, int intraSplit,
                      bool intra, int partMode, int parent_cb, int parent_cr, int cbf_pct)
  {
    const seq_parameter_set& S = *sps;
    int split;
    if (log2 <= S.Log2MaxTrafoSize && log2 > S.Log2MinTrafoSize && depth < maxDepth && !(intraSplit && depth == 0)) {
      split = pct(30);
      bit(CONTEXT_MODEL_SPLIT_TRANSFORM_FLAG + 5 - log2, split);
    } else {
      const int interSplit = S.max_transform_hierarchy_depth_inter == 0 && depth == 0 && !intra && partMode != PART_2Nx2N;
      split = log2 > S.Log2MaxTrafoSize || (intraSplit && depth == 0) || interSplit;
    }
    int cbf_cb = -1, cbf_cr = -1;
    if (log2 > 2) {
      if (parent_cb) { cbf_cb = pct(cbf_pct); bit(CONTEXT_MODEL_CBF_CHROMA + depth, cbf_cb); }
      if (parent_cr) { cbf_cr = pct(cbf_pct); bit(CONTEXT_MODEL_CBF_CHROMA + depth, cbf_cr); }
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
    /* transform unit: no cu_qp_delta (pps.cu_qp_delta_enabled_flag = 0), no cross-component prediction */
    if (cbf_luma) residual(x0, y0, log2, 0, intra);
    if (log2 > 2) {
      if (cbf_cb) residual(x0, y0, log2 - 1, 1, intra);
      if (cbf_cr) residual(x0, y0, log2 - 1, 2, intra);
    } else if (blkIdx == 3) {
      if (cbf_cb) residual(xBase, yBase, 2, 1, intra);
      if (cbf_cr) residual(xBase, yBase, 2, 2, intra);
    }
  }

  /* ---------------- 7.3.8.5 coding unit (read_coding_unit, slice.cc:4315-4636) ---------------- */
  void coding_unit(int x0, int y0, int log2, int ctDepth)
  {
    const seq_parameter_set& S = *sps;
    const int nCbS = 1 << log2;
    img.set_log2CbSize(x0, y0, log2, true);
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
      bit(CONTEXT_MODEL_PART_MODE + 0, part == PART_2Nx2N);
      if (part != PART_2Nx2N) {
        const bool horiz = part == PART_2NxN || part == PART_2NxnU || part == PART_2NxnD;
        bit(CONTEXT_MODEL_PART_MODE + 1, horiz);
        if (log2 > S.Log2MinCbSizeY && S.amp_enabled_flag) {
          const bool plain = part == PART_2NxN || part == PART_Nx2N;
          bit(CONTEXT_MODEL_PART_MODE + 3, plain);
          if (!plain) bypass(part == PART_2NxnD || part == PART_nRx2N);
        }
        /* log2 == min (8): bit1 = 1 -> 2NxN, 0 -> Nx2N, nothing else to code (slice.cc:1788-1800) */
      }
    }
    img.set_PartMode(x0, y0, (enum PartMode)part);
    bool merge_2Nx2N = false;
    if (intra) {
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
      const int icpm = below(5);                                               /* intra_chroma_pred_mode, 4 = derived from luma */
      bit(CONTEXT_MODEL_INTRA_CHROMA_PRED_MODE, icpm != 4);
      if (icpm != 4) cabac->write_CABAC_FL_bypass(icpm, 2);
      const int lumaMode = img.get_IntraPredMode(x0, y0);
      int modeC;
      if (icpm == 4) modeC = lumaMode;
      else { static const int tab[4] = {0, 26, 10, 1}; modeC = tab[icpm]; if (modeC == lumaMode) modeC = 34; }
      img.set_IntraPredModeC(x0, y0, log2, (enum IntraPredMode)modeC, icpm == 4);
    } else {
      const int q = nCbS / 4, h = nCbS / 2;
      switch (part) {
        case PART_2Nx2N: merge_2Nx2N = prediction_unit(nCbS, nCbS, ctDepth); break;
        case PART_2NxN: prediction_unit(nCbS, h, ctDepth); prediction_unit(nCbS, h, ctDepth); break;
        case PART_Nx2N: prediction_unit(h, nCbS, ctDepth); prediction_unit(h, nCbS, ctDepth); break;
        case PART_2NxnU: prediction_unit(nCbS, q, ctDepth); prediction_unit(nCbS, nCbS - q, ctDepth); break;
        case PART_2NxnD: prediction_unit(nCbS, nCbS - q, ctDepth); prediction_unit(nCbS, q, ctDepth); break;
        case PART_nLx2N: prediction_unit(q, nCbS, ctDepth); prediction_unit(nCbS - q, nCbS, ctDepth); break;
        default: prediction_unit(nCbS - q, nCbS, ctDepth); prediction_unit(q, nCbS, ctDepth); break;
      }
    }
    bool root_cbf = true;
    if (!intra && !(part == PART_2Nx2N && merge_2Nx2N)) { root_cbf = pct(70); bit(CONTEXT_MODEL_RQT_ROOT_CBF, root_cbf); }
    if (root_cbf) {
      const int maxDepth = intra ? S.max_transform_hierarchy_depth_intra + intraSplit : S.max_transform_hierarchy_depth_inter;
      transform_tree(x0, y0, x0, y0, log2, 0, 0, maxDepth, intraSplit, intra, part, 1, 1, intra ? 60 : 45);
    }
  }

  /* ---------------- one picture: NAL header + slice header + slice data (tiles = CABAC substreams) ---------------- */
  void slice_data(CABAC_encoder_bitstream& enc, uint32_t seed, std::vector<int>& substream_end)
  {
    cabac = &enc;
    s = seed ? seed : 1;
    img.clear_metadata();
    const seq_parameter_set& S = *sps;
    const int nCtb = S.PicSizeInCtbsY, W = S.PicWidthInCtbsY;
    enc.set_context_models(&ctx);
    ctx.init(shdr->initType, shdr->SliceQPY);
    enc.init_CABAC();
    const int start = enc.size();
    substream_end.clear();
    for (int ts = 0; ts < nCtb; ts++) {
      const int rs = pps->scan->CtbAddrTStoRS[ts], xCtb = rs % W, yCtb = rs / W;
      img.set_SliceAddrRS(xCtb, yCtb, shdr->SliceAddrRS);
      img.set_SliceHeaderIndex(xCtb << S.Log2CtbSizeY, yCtb << S.Log2CtbSizeY, 0);
      if (shdr->slice_sao_luma_flag || shdr->slice_sao_chroma_flag) write_sao(xCtb, yCtb, rs);
      const int target = 3 + below(S.Log2CtbSizeY - 2);
      coding_quadtree(xCtb << S.Log2CtbSizeY, yCtb << S.Log2CtbSizeY, S.Log2CtbSizeY, 0, target);
      const bool last = ts == nCtb - 1;
      enc.write_CABAC_term_bit(last);                                          /* end_of_slice_segment_flag */
      if (!last && pps->scan->TileId[ts + 1] != pps->scan->TileId[ts]) {
        enc.write_CABAC_term_bit(1);                                           /* end_of_subset_one_bit */
        enc.flush_CABAC();
        enc.add_trailing_bits();                                               /* byte_alignment() */
        enc.flush_VLC();
        substream_end.push_back(enc.size() - start);
        ctx.init(shdr->initType, shdr->SliceQPY);
        enc.init_CABAC();
      }
    }
    enc.flush_CABAC();
    enc.add_trailing_bits();                                                   /* rbsp_slice_segment_trailing_bits */
    enc.flush_VLC();
  }
};

int run(const Cfg& cfg, const char* out_name)
{
  de265_init();                                /* scan-order tables (scan.cc) used by the residual writer */
  std::unique_ptr<Gen> G(new Gen);
  Gen& g = *G;
  g.cfg = cfg;
  g.vps = std::make_shared<video_parameter_set>();
  g.sps = std::make_shared<seq_parameter_set>();
  g.pps = std::make_shared<pic_parameter_set>();
  g.vps->set_defaults(Profile_Main, 6, 2);
  seq_parameter_set& S = *g.sps;
  S.set_defaults();
  S.set_CB_log2size_range(3, 6);
  S.set_TB_log2size_range(2, 5);
  S.set_resolution(cfg.W, cfg.H);
  S.chroma_format_idc = 1;
  S.bit_depth_luma = S.bit_depth_chroma = cfg.bd;
  S.log2_max_pic_order_cnt_lsb = 8;
  S.sps_max_dec_pic_buffering[0] = 4; S.sps_max_num_reorder_pics[0] = 0; S.sps_max_latency_increase_plus1[0] = 0;
  S.max_transform_hierarchy_depth_inter = 2;
  S.max_transform_hierarchy_depth_intra = 2;
  S.amp_enabled_flag = 1;
  S.sample_adaptive_offset_enabled_flag = cfg.sao ? 1 : 0;
  S.pcm_enabled_flag = 0;
  S.long_term_ref_pics_present_flag = 0;
  S.sps_temporal_mvp_enabled_flag = 0;
  S.strong_intra_smoothing_enable_flag = 1;
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
  P.loop_filter_across_tiles_enabled_flag = 1;
  P.pps_loop_filter_across_slices_enabled_flag = 1;
  P.deblocking_filter_control_present_flag = 0;
  P.pic_cb_qp_offset = 1; P.pic_cr_qp_offset = -1;
  P.set_derived_values(g.sps.get());
  P.pps_read = true;

  if (g.img.alloc_image(cfg.W, cfg.H, de265_chroma_420, g.sps, true, &g.dctx, 0, nullptr, false) != DE265_OK) return 2;
  g.img.set_headers(g.vps, g.sps, g.pps);
  g.ectx.img = &g.img;

  FILE* f = fopen(out_name, "wb");
  if (!f) return 2;
  CABAC_encoder_bitstream out;
  nal_header nal;
  out.write_startcode(); nal.set(NAL_UNIT_VPS_NUT); nal.write(out); g.vps->write(&g.dctx, out); out.add_trailing_bits(); out.flush_VLC();
  out.write_startcode(); nal.set(NAL_UNIT_SPS_NUT); nal.write(out); g.sps->write(&g.dctx, out); out.add_trailing_bits(); out.flush_VLC();
  out.write_startcode(); nal.set(NAL_UNIT_PPS_NUT); nal.write(out); g.pps->write(&g.dctx, out, g.sps.get()); out.add_trailing_bits(); out.flush_VLC();
  fwrite(out.data(), 1, out.size(), f);

  for (int fr = 0; fr < cfg.frames; fr++) {
    slice_segment_header* sh = new slice_segment_header;                      /* owned by the image (img.slices) */
    g.shdr = sh;
    g.img.slices.clear();
    g.img.slices.push_back(sh);
    const int type = fr == 0 ? SLICE_TYPE_I : ((cfg.b_frames && (fr & 1) == 0) ? SLICE_TYPE_B : SLICE_TYPE_P);
    const int nal_type = fr == 0 ? NAL_UNIT_IDR_W_RADL : NAL_UNIT_TRAIL_R;
    const int nrefs = fr >= 2 ? 2 : fr;
    sh->first_slice_segment_in_pic_flag = 1;
    sh->slice_pic_parameter_set_id = 0;
    sh->slice_type = type;
    sh->pic_output_flag = 1;
    sh->slice_pic_order_cnt_lsb = fr & 0xFF;
    sh->short_term_ref_pic_set_sps_flag = 1;
    sh->short_term_ref_pic_set_idx = nrefs >= 2 ? 1 : 0;
    sh->slice_sao_luma_flag = sh->slice_sao_chroma_flag = cfg.sao ? 1 : 0;
    sh->num_ref_idx_active_override_flag = type != SLICE_TYPE_I;
    sh->num_ref_idx_l0_active = nrefs; sh->num_ref_idx_l1_active = nrefs;
    sh->five_minus_max_num_merge_cand = 0;
    sh->slice_qp_delta = 0;
    sh->slice_loop_filter_across_slices_enabled_flag = 1;
    sh->slice_deblocking_filter_disabled_flag = P.pic_disable_deblocking_filter_flag;
    sh->pps = g.pps;
    sh->SliceAddrRS = 0; sh->slice_segment_address = 0;
    sh->compute_derived_values(g.pps.get());
    g.slice_type = type; g.nref[0] = g.nref[1] = nrefs;
    const uint32_t seed = cfg.seed * 2654435761u + 977u * (uint32_t)fr + 1u;

    /* pass 1: the slice data alone, for the entry points of the tile substreams */
    std::vector<int> ends;
    { CABAC_encoder_bitstream scratch; g.slice_data(scratch, seed, ends); }
    sh->num_entry_point_offsets = (int)ends.size();
    sh->entry_point_offset.clear();
    int maxd = 1;
    for (size_t i = 0; i < ends.size(); i++) { const int d = ends[i] - (i ? ends[i - 1] : 0); if (d > maxd) maxd = d; sh->entry_point_offset.push_back((uint32_t)ends[i]); }
    sh->offset_len = 1; while ((1 << sh->offset_len) < maxd) sh->offset_len++;
    /* pass 2: NAL header, slice header, the same slice data */
    CABAC_encoder_bitstream enc;
    enc.write_startcode();
    nal.set(nal_type); nal.write(enc);
    /* the header writer takes num_ref_idx_lX_active as the syntax element (minus 1) and leaves the count behind */
    sh->num_ref_idx_l0_active = nrefs - 1; sh->num_ref_idx_l1_active = nrefs - 1;
    if (sh->write(&g.dctx, enc, g.sps.get(), g.pps.get(), (uint8_t)nal_type) != DE265_OK) { fprintf(stderr, "streamgen: slice header not writable\n"); return 2; }
    if (type == SLICE_TYPE_I) { sh->num_ref_idx_l0_active = sh->num_ref_idx_l1_active = 0; }
    if (type == SLICE_TYPE_P) sh->num_ref_idx_l1_active = 0;
    enc.add_trailing_bits();                                                   /* byte_alignment() of the slice header */
    enc.flush_VLC();
    std::vector<int> ends2;
    g.slice_data(enc, seed, ends2);
    if (ends2 != ends) { fprintf(stderr, "streamgen: substream sizes changed between the passes\n"); return 3; }
    fwrite(enc.data(), 1, enc.size(), f);
    g.img.slices.clear();
    delete sh;
  }
  fclose(f);
  return 0;
}

} // namespace

int main(int argc, char** argv)
{
  if (argc < 9) { fprintf(stderr, "usage: %s out.h265 W H bit_depth tile_cols tile_rows n_frames seed [intra_pct=5] [b_frames=1] [sao=1]\n", argv[0]); return 2; }
  Cfg c;
  c.W = atoi(argv[2]); c.H = atoi(argv[3]); c.bd = atoi(argv[4]); c.tc = atoi(argv[5]); c.tr = atoi(argv[6]); c.frames = atoi(argv[7]);
  c.seed = (uint32_t)strtoul(argv[8], nullptr, 0);
  c.intra_pct = argc > 9 ? atoi(argv[9]) : 5; c.b_frames = argc > 10 ? atoi(argv[10]) : 1; c.sao = argc > 11 ? atoi(argv[11]) : 1;
  if (c.W % 8 || c.H % 8 || c.W < 16 || c.H < 16 || c.bd < 8 || c.bd > 12 || c.tc < 1 || c.tr < 1 || c.frames < 1) { fprintf(stderr, "streamgen: bad arguments\n"); return 2; }
  return run(c, argv[1]);
}
