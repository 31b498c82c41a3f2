/*
 * ref_replay.cc — TEST INFRASTRUCTURE (linked into oracle/_ref/libde265_ref.so only; never into the product).
 *
 * Replays one picture's work lists (include/de265_mi355x.h) through the REAL reference code, compiled from
 * /root/reference where it lies: the picture is rebuilt as a de265_image with the reference's own metadata
 * arrays (cb_info / tu_info / pb_info / ctb_info, image.h:389-395), SPS / PPS / slice headers, and then the
 * reference's own functions do the pixel work:
 *     generate_inter_prediction_samples   motion.cc:288      (mc_luma / mc_chroma incl. edge padding, weighted pred.)
 *     scale_coefficients                  transform.cc:645   (dequant + IDCT / IDST / skip / bypass + add)
 *     decode_intra_prediction             intrapred.cc:321   (border construction, substitution, filters, predictors)
 *     apply_deblocking_filter             deblock.cc:908     (edge flags, bS, luma + chroma filters)
 *     apply_sample_adaptive_offset_sequential  sao.cc:327
 * with either the scalar fallback table or the SSE/AVX tables (accel != 0).  It is the second pin of the CPU
 * oracle (oracle/hevc_oracle.c): the first is the recorded girlshy stream; this one covers what girlshy does
 * not exercise — tiles, 10/12-bit, explicit weights, out-of-picture MVs, transform skip, every CU/TU size —
 * on the synthetic pictures the GPU parity tests use.  It is also the "reference" CPU baseline of bench.py.
 *
 * Missing-reference fills (M355_PBF_FILL_*): the block's DPB slot holds no image here either, so the reference's own
 * error path (motion.cc:362-376) produces the 1 << 13 prediction.  Blocks that carry already-scaled levels
 * (M355_RBF_DEQUANTIZED, what the table slots receive) enter the reference below its dequantiser: the levels go into
 * coeffBuf and the reference's own slots (transform_add / transform_skip_residual / rdpcm_* + add_residual) run on them,
 * as scale_coefficients_internal calls them (transform.cc:530-625).  PCM blocks: the sample store is done here
 * (read_pcm_samples consumes the bitstream), the filters' treatment of PCM units is the reference's.
 */
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <chrono>
#include <memory>
#include <thread>
#include <unordered_map>
#include <vector>

#include "libde265/de265.h"
#include "libde265/deblock.h"
#include "libde265/decctx.h"
#include "libde265/image.h"
#include "libde265/intrapred.h"
#include "libde265/motion.h"
#include "libde265/pps.h"
#include "libde265/sao.h"
#include "libde265/slice.h"
#include "libde265/sps.h"
#include "libde265/transform.h"

#include "de265_mi355x.h"

namespace {

/* motion compensation asks its context for the reference pictures by frame id (motion.cc:340) */
struct ReplayContext : public base_context {
  std::vector<de265_image*> frames;
  const de265_image* get_image(uint16_t id) const override { return id < frames.size() ? frames[id] : nullptr; }
  bool has_image(uint16_t id) const override { return id < frames.size() && frames[id] != nullptr; }
};

void import_plane(de265_image* img, int c, const void* src)
{
  const int w = img->get_width(c), h = img->get_height(c), bpp = img->get_bytes_per_pixel(c);
  const ptrdiff_t stride = img->get_image_stride(c);
  for (int y = 0; y < h; y++)
    memcpy(img->get_image_plane(c) + (size_t)y * stride * bpp, (const uint8_t*)src + (size_t)y * w * bpp, (size_t)w * bpp);
}
void export_plane(const de265_image* img, int c, void* dst)
{
  const int w = img->get_width(c), h = img->get_height(c), bpp = img->get_bytes_per_pixel(c);
  const ptrdiff_t stride = img->get_image_stride(c);
  for (int y = 0; y < h; y++)
    memcpy((uint8_t*)dst + (size_t)y * w * bpp, img->get_image_plane(c) + (size_t)y * stride * bpp, (size_t)w * bpp);
}

int ilog2(int v) { int r = 0; while ((1 << r) < v) r++; return r; }

} // namespace

extern "C" {

/* ref_planes: M355_MAX_REF_FRAMES * 3 pointers ([slot * 3 + c], tight w x h arrays of uint8 / uint16) or NULL;
 * out_planes: 3 tight arrays receiving the replayed picture.  stages: M355_STAGE_* mask.  accel: 0 = scalar
 * fallback table (init_acceleration_functions_fallback), 1 = de265_acceleration_AUTO (SSE4.1 / AVX2 / AVX-512).
 * returns 0, or a negative code: -1 unsupported list content, -2 reference error (allocation, derived values). */
__attribute__((visibility("default")))
int m355_ref_replay(const m355_picture* pic, const void* const* ref_planes, int stages, int accel, void* const* out_planes)
{
  const m355_pic_params& pp = pic->pp;

  decoder_context dctx;                          /* img->decctx: acceleration table for intra / deblock / transforms */
  ReplayContext rctx;                            /* the base_context motion compensation sees */
  dctx.set_acceleration_functions(accel ? de265_acceleration_AUTO : de265_acceleration_SCALAR);
  rctx.set_acceleration_functions(accel ? de265_acceleration_AUTO : de265_acceleration_SCALAR);

  /* ---- SPS / PPS from the picture parameters ---- */
  std::shared_ptr<seq_parameter_set> sps = std::make_shared<seq_parameter_set>();
  sps->set_defaults();
  sps->chroma_format_idc = pp.chroma_format_idc;
  sps->pic_width_in_luma_samples = pp.width; sps->pic_height_in_luma_samples = pp.height;
  sps->bit_depth_luma = pp.bit_depth_luma; sps->bit_depth_chroma = pp.bit_depth_chroma;
  sps->log2_min_luma_coding_block_size = pp.log2_min_cb_size;
  sps->log2_diff_max_min_luma_coding_block_size = pp.log2_ctb_size - pp.log2_min_cb_size;
  sps->log2_min_transform_block_size = pp.log2_min_tb_size;
  sps->log2_diff_max_min_transform_block_size = (pp.log2_ctb_size < 5 ? pp.log2_ctb_size : 5) - pp.log2_min_tb_size;
  sps->max_transform_hierarchy_depth_inter = pp.log2_ctb_size - pp.log2_min_tb_size;
  sps->max_transform_hierarchy_depth_intra = pp.log2_ctb_size - pp.log2_min_tb_size;
  sps->amp_enabled_flag = 1;
  sps->sample_adaptive_offset_enabled_flag = (pp.flags & M355_PF_SAO_ENABLED) ? 1 : 0;
  sps->pcm_loop_filter_disable_flag = (pp.flags & M355_PF_PCM_LOOP_FILTER_DISABLE) ? 1 : 0;
  sps->strong_intra_smoothing_enable_flag = (pp.flags & M355_PF_STRONG_INTRA_SMOOTHING) ? 1 : 0;
  sps->range_extension.intra_smoothing_disabled_flag = (pp.flags & M355_PF_INTRA_SMOOTHING_DISABLED) ? 1 : 0;
  sps->range_extension.implicit_rdpcm_enabled_flag = (pp.flags & M355_PF_IMPLICIT_RDPCM) ? 1 : 0;
  sps->range_extension.transform_skip_rotation_enabled_flag = (pp.flags & M355_PF_TRANSFORM_SKIP_ROTATION) ? 1 : 0;
  if (pp.flags & M355_PF_SCALING_LIST) {
    /* ScalingFactor tables as the lists carry them: [sizeId][matrixID][y][x] (sps.h:58-65) */
    sps->scaling_list_enable_flag = 1;
    const uint8_t* f = pic->scaling_factors;
    memcpy(sps->scaling_list.ScalingFactor_Size0, f, 6 * 16); f += 6 * 16;
    memcpy(sps->scaling_list.ScalingFactor_Size1, f, 6 * 64); f += 6 * 64;
    memcpy(sps->scaling_list.ScalingFactor_Size2, f, 6 * 256); f += 6 * 256;
    memcpy(sps->scaling_list.ScalingFactor_Size3, f, 6 * 1024);
  }
  if (sps->compute_derived_values(false) != DE265_OK) return -2;
  sps->sps_read = true;

  std::shared_ptr<pic_parameter_set> pps[2];     /* [weighted prediction off / on]: the PB's own flag selects the branch */
  for (int k = 0; k < 2; k++) {
    pps[k] = std::make_shared<pic_parameter_set>();
    pic_parameter_set& q = *pps[k];
    q.set_defaults();
    q.sps = sps;
    q.constrained_intra_pred_flag = (pp.flags & M355_PF_CONSTRAINED_INTRA_PRED) ? 1 : 0;
    q.transform_skip_enabled_flag = 1;
    q.pic_cb_qp_offset = pp.pic_cb_qp_offset; q.pic_cr_qp_offset = pp.pic_cr_qp_offset;
    q.weighted_pred_flag = k; q.weighted_bipred_flag = k;
    q.transquant_bypass_enable_flag = 1;
    q.tiles_enabled_flag = (pp.num_tile_cols > 1 || pp.num_tile_rows > 1) ? 1 : 0;
    q.num_tile_columns = pp.num_tile_cols; q.num_tile_rows = pp.num_tile_rows;
    q.uniform_spacing_flag = 0;
    for (int i = 0; i < pp.num_tile_cols; i++) q.colWidth[i] = pp.col_bd[i + 1] - pp.col_bd[i];
    for (int i = 0; i < pp.num_tile_rows; i++) q.rowHeight[i] = pp.row_bd[i + 1] - pp.row_bd[i];
    q.loop_filter_across_tiles_enabled_flag = (pp.flags & M355_PF_LF_ACROSS_TILES) ? 1 : 0;
    q.deblocking_filter_control_present_flag = 1;
    q.range_extension.cross_component_prediction_enabled_flag = (pp.flags & M355_PF_CROSS_COMPONENT_PRED) != 0;
    q.scaling_list = sps->scaling_list;          /* transform.cc:505-508 reads the PPS copy (pps.cc copies the SPS list when it has none) */
    q.set_derived_values(sps.get());
    q.pps_read = true;
  }

  /* ---- images: references and the picture ---- */
  const de265_chroma chroma = (de265_chroma)pp.chroma_format_idc;
  const int nc = pp.chroma_format_idc ? 3 : 1;
  std::vector<std::unique_ptr<de265_image>> owned;
  rctx.frames.assign(M355_MAX_REF_FRAMES, nullptr);
  for (int s = 0; s < M355_MAX_REF_FRAMES; s++) {
    if (!ref_planes || !ref_planes[s * 3]) continue;
    owned.emplace_back(new de265_image);
    de265_image* r = owned.back().get();
    if (r->alloc_image(pp.width, pp.height, chroma, sps, false, &dctx, 0, nullptr, false) != DE265_OK) return -2;
    for (int c = 0; c < nc; c++) import_plane(r, c, ref_planes[s * 3 + c]);
    r->PicState = UsedForShortTermReference;
    rctx.frames[s] = r;
  }
  owned.emplace_back(new de265_image);
  de265_image* img = owned.back().get();
  if (img->alloc_image(pp.width, pp.height, chroma, sps, true, &dctx, 0, nullptr, false) != DE265_OK) return -2;
  img->set_headers(nullptr, sps, pps[0]);
  img->clear_metadata();
  img->fill_image(0, 0, 0);                      /* planes are zero at allocation in the decoder (image.cc:164) */

  /* ---- slice headers ---- */
  for (int i = 0; i < pic->n_slices; i++) {
    const m355_slice& sl = pic->slices[i];
    slice_segment_header* sh = new slice_segment_header;      /* owned (and deleted) by the image */
    sh->slice_index = i;
    sh->pps = pps[0];
    sh->slice_type = SLICE_TYPE_B;
    sh->SliceAddrRS = sl.slice_addr_rs;
    sh->slice_segment_address = sl.slice_addr_rs;
    sh->slice_deblocking_filter_disabled_flag = (sl.flags & M355_SF_DEBLOCK_DISABLED) != 0;
    sh->slice_beta_offset = sl.beta_offset; sh->slice_tc_offset = sl.tc_offset;
    sh->slice_loop_filter_across_slices_enabled_flag = (sl.flags & M355_SF_LF_ACROSS_SLICES) != 0;
    sh->slice_sao_luma_flag = (sl.flags & M355_SF_SAO_LUMA) != 0;
    sh->slice_sao_chroma_flag = (sl.flags & M355_SF_SAO_CHROMA) != 0;
    for (int l = 0; l < 2; l++)
      for (int k = 0; k < MAX_NUM_REF_PICS; k++) sh->RefPicList[l][k] = k;     /* refIdx == DPB slot */
    img->slices.push_back(sh);
  }

  /* ---- metadata: CTBs, CUs, transform tree, motion ---- */
  const int cs = 1 << pp.log2_ctb_size;
  const int ctbW = (pp.width + cs - 1) / cs, ctbH = (pp.height + cs - 1) / cs;
  for (int cy = 0; cy < ctbH; cy++)
    for (int cx = 0; cx < ctbW; cx++) {
      const m355_ctb& cb = pic->ctbs[cy * ctbW + cx];
      img->set_SliceHeaderIndex(cx * cs, cy * cs, cb.slice_idx);
      img->set_SliceAddrRS(cx, cy, pic->slices[cb.slice_idx].slice_addr_rs);
      sao_info si;
      si.SaoTypeIdx = cb.sao_type; si.SaoEoClass = cb.sao_eo_class;
      for (int k = 0; k < 3; k++) {
        si.sao_band_position[k] = cb.sao_band_pos[k];
        for (int j = 0; j < 4; j++) si.saoOffsetVal[k][j] = cb.sao_offset[k][j];
      }
      img->set_sao_info(cx, cy, &si);
    }
  for (int i = 0; i < pic->n_cus; i++) {
    const m355_cu& cu = pic->cus[i];
    img->set_log2CbSize(cu.x, cu.y, cu.log2_size, true);
    img->set_pred_mode(cu.x, cu.y, cu.log2_size, (PredMode)cu.pred_mode);
    img->set_PartMode(cu.x, cu.y, (PartMode)cu.part_mode);
    img->set_QPY(cu.x, cu.y, cu.log2_size, cu.qp_y);
    if (cu.flags & M355_CUF_PCM) img->set_pcm_flag(cu.x, cu.y, cu.log2_size);
    if (cu.flags & M355_CUF_TRANSQUANT_BYPASS) img->set_cu_transquant_bypass(cu.x, cu.y, cu.log2_size);
  }
  for (int i = 0; i < pic->n_tus; i++) {
    /* a transform-tree leaf of size s inside a CU of size c sits at depth c - s: every ancestor is split */
    const m355_tu& tu = pic->tus[i];
    const int cl2 = img->get_log2CbSize(tu.x, tu.y);
    for (int d = 0; d < cl2 - tu.log2_size; d++) {
      const int a = cl2 - d;                       /* log2 size of the ancestor at depth d */
      img->set_split_transform_flag((tu.x >> a) << a, (tu.y >> a) << a, d);
    }
    if (tu.flags & M355_TUF_NONZERO_COEFF) img->set_nonzero_coefficient(tu.x, tu.y, tu.log2_size);
  }
  for (int i = 0; i < pic->n_pbs; i++) {
    const m355_pb& pb = pic->pbs[i];
    PBMotion mv;
    memset(&mv, 0, sizeof(mv));
    for (int l = 0; l < 2; l++) {
      mv.predFlag[l] = (pb.flags & (M355_PBF_PRED_L0 << l)) ? 1 : 0;
      if (mv.predFlag[l]) {
        if (pb.ref_slot[l] < 0 || pb.ref_slot[l] >= MAX_NUM_REF_PICS) return -1;
        if ((pb.flags & (M355_PBF_FILL_L0 << l)) && rctx.frames[pb.ref_slot[l]]) return -1;   /* a fill must name a slot without a picture */
        mv.refIdx[l] = (uint8_t)pb.ref_slot[l];
        mv.mv[l].x = pb.mv[l][0]; mv.mv[l].y = pb.mv[l][1];
      }
    }
    img->set_mv_info(pb.x, pb.y, pb.w, pb.h, mv);
  }

  /* ---- inter prediction ---- */
  if (stages & M355_STAGE_INTER) {
    const int shift1_l = 14 - pp.bit_depth_luma, shift1_c = 14 - pp.bit_depth_chroma;
    const int osl = pp.bit_depth_luma - 8, osc = pp.bit_depth_chroma - 8;     /* WpOffsetBdShift (sps.cc:639-640) */
    for (int i = 0; i < pic->n_pbs; i++) {
      const m355_pb& pb = pic->pbs[i];
      slice_segment_header* sh = img->get_SliceHeader(pb.x, pb.y);
      const bool weighted = (pb.flags & M355_PBF_WEIGHTED) != 0;
      sh->pps = pps[weighted ? 1 : 0];
      PBMotion mv = img->get_mv_info(pb.x, pb.y);
      if (weighted) {
        /* this PB's explicit weights become the slice's table entries for the reference indices it uses
           (pred_weight_table semantics, slice.cc:159-231; offsets are stored unshifted there) */
        for (int l = 0; l < 2; l++) {
          if (!mv.predFlag[l]) continue;
          const m355_wt& w = pic->wts[pb.wt_idx[l]];
          const int r = mv.refIdx[l];
          sh->luma_log2_weight_denom = (uint8_t)(w.log2wd_luma - (shift1_l < 2 ? 2 : shift1_l));
          sh->ChromaLog2WeightDenom = (uint8_t)(w.log2wd_chroma - (shift1_c < 2 ? 2 : shift1_c));
          sh->LumaWeight[l][r] = w.w[0]; sh->luma_offset[l][r] = (int16_t)(w.o[0] >> osl);
          for (int k = 0; k < 2; k++) { sh->ChromaWeight[l][r][k] = w.w[1 + k]; sh->ChromaOffset[l][r][k] = (int16_t)(w.o[1 + k] >> osc); }
        }
      }
      generate_inter_prediction_samples(&rctx, sh, img, pb.x, pb.y, 0, 0, 64, pb.w, pb.h, &mv);
      sh->pps = pps[0];
    }
  }

  /* ---- residuals (and intra prediction, which interleaves with its blocks' residuals in decode order) ---- */
  std::unique_ptr<thread_context> tctx(new thread_context);
  tctx->decctx = &dctx; tctx->img = img;
  memset(tctx->coeffBuf, 0, 32 * 32 * sizeof(int16_t));
  const int sw = (pp.chroma_format_idc == 1 || pp.chroma_format_idc == 2) ? 2 : 1, sh_ = pp.chroma_format_idc == 1 ? 2 : 1;
  auto run_rb = [&](const m355_rb& rb) -> bool {
    const int nT = 1 << rb.log2_size, c = rb.cidx;
    const int xl = c ? rb.x * sw : rb.x, yl = c ? rb.y * sh_ : rb.y;
    tctx->shdr = img->get_SliceHeader(xl, yl);
    tctx->nCoeff[c] = rb.ncoeff;
    for (int k = 0; k < rb.ncoeff; k++) {
      const uint32_t v = pic->coeffs[rb.coeff_ofs + k];
      tctx->coeffPos[c][k] = (int16_t)(v & 0xFFFF); tctx->coeffList[c][k] = (int16_t)(v >> 16);
    }
    tctx->qPYPrime = tctx->qPCbPrime = tctx->qPCrPrime = rb.qp;
    tctx->cu_transquant_bypass_flag = rb.kind == M355_RK_BYPASS;
    tctx->ResScaleVal = 0;
    if ((pp.flags & M355_PF_CROSS_COMPONENT_PRED) && c && ((rb.matrix_id >> 4) & 7)) {
      /* the TU's luma block ran just before (list / decode order), so tctx->residual_luma is this TU's */
      const int v = (rb.matrix_id >> 4) & 7;
      tctx->ResScaleVal = (rb.matrix_id & 0x80) ? -(1 << (v - 1)) : (1 << (v - 1));
    }
    const int rdpcm = (rb.flags & M355_RBF_RDPCM_H) ? 1 : ((rb.flags & M355_RBF_RDPCM_V) ? 2 : 0);
    const bool intra = img->get_pred_mode(xl, yl) == MODE_INTRA;
    if ((rb.flags & M355_RBF_DEQUANTIZED) && rb.kind != M355_RK_BYPASS) {
      /* below the dequantiser: the calls of transform.cc:530-625 on the block's levels as they are */
      if (pp.flags & M355_PF_CROSS_COMPONENT_PRED) return false;
      acceleration_functions& acc = dctx.acceleration;
      const int bd = c ? pp.bit_depth_chroma : pp.bit_depth_luma;
      int16_t* coeff = tctx->coeffBuf;
      for (int k = 0; k < rb.ncoeff; k++) coeff[tctx->coeffPos[c][k]] = tctx->coeffList[c][k];
      uint8_t* pred = img->get_image_plane(c) + ((size_t)rb.y * img->get_image_stride(c) + rb.x) * img->get_bytes_per_pixel(c);
      const int stride = img->get_image_stride(c);
      if (rb.kind == M355_RK_SKIP) {
        const int bdShift = 20 - bd, tsShift = 5 + rb.log2_size;
        if (rb.flags & M355_RBF_ROTATE) acc.rotate_coefficients(coeff, nT);
        int32_t residual[32 * 32];
        if (rdpcm == 2) acc.rdpcm_v(residual, coeff, nT, tsShift, bdShift);
        else if (rdpcm == 1) acc.rdpcm_h(residual, coeff, nT, tsShift, bdShift);
        else acc.transform_skip_residual(residual, coeff, nT, tsShift, bdShift);
        if (bd > 8) acc.add_residual<uint16_t>((uint16_t*)pred, stride, residual, nT, bd); else acc.add_residual<uint8_t>(pred, stride, residual, nT, bd);
      } else if (rb.kind == M355_RK_DST) {
        if (bd > 8) acc.transform_4x4_dst_add<uint16_t>((uint16_t*)pred, coeff, stride, bd); else acc.transform_4x4_dst_add<uint8_t>(pred, coeff, stride, bd);
      } else {
        if (bd > 8) acc.transform_add<uint16_t>(rb.log2_size - 2, (uint16_t*)pred, coeff, stride, bd); else acc.transform_add<uint8_t>(rb.log2_size - 2, pred, coeff, stride, bd);
      }
      memset(coeff, 0, 32 * 32 * sizeof(int16_t));
      return true;
    }
    scale_coefficients(tctx.get(), rb.x, rb.y, rb.x, rb.y, nT, c, rb.kind == M355_RK_SKIP, intra, rdpcm);
    return true;
  };
  const int nrb = pic->rb_count[0] + pic->rb_count[1] + pic->rb_count[2] + pic->rb_count[3];
  std::unordered_map<uint32_t, int> deferred;    /* res_ofs -> rb index */
  for (int i = 0; i < nrb; i++) {
    const m355_rb& rb = pic->rbs[i];
    if (rb.flags & M355_RBF_DEFERRED) deferred[rb.res_ofs] = i;
    else if (stages & M355_STAGE_RESIDUAL) { if (!run_rb(rb)) return -1; }
  }
  if (stages & M355_STAGE_INTRA) {
    /* decode order = tile scan over the CTBs, each CTB's blocks as listed */
    const pic_parameter_set& q = *pps[0];
    for (int ts = 0; ts < ctbW * ctbH; ts++) {
      const m355_ctb& cb = pic->ctbs[q.scan->CtbAddrTStoRS[ts]];
      for (uint32_t k = 0; k < cb.ib_count; k++) {
        const m355_ib& ib = pic->ibs[cb.ib_start + k];
        if (ib.flags & M355_IBF_PCM) {
          /* read_pcm_samples (slice.cc:4211-4255) consumes the bitstream and stores sample << (BitDepth - PcmBitDepth);
             the lists carry the stored values, so the store is all that is left to do here — what this pins is the
             REFERENCE's treatment of PCM units in the filters (pcm_loop_filter_disable, deblock.cc:576-592, sao.cc:103-120) */
          const int n = 1 << ib.log2_size, bpp = img->get_bytes_per_pixel(ib.cidx);
          const ptrdiff_t st = img->get_image_stride(ib.cidx);
          for (int y = 0; y < n; y++)
            for (int x = 0; x < n; x++) {
              const uint16_t v = pic->pcm[ib.res_ofs + y * n + x];
              uint8_t* q = img->get_image_plane(ib.cidx) + ((size_t)(ib.y + y) * st + ib.x + x) * bpp;
              if (bpp == 1) *q = (uint8_t)v; else *(uint16_t*)q = v;
            }
          continue;
        }
        decode_intra_prediction(img, ib.x, ib.y, (IntraPredMode)ib.mode, 1 << ib.log2_size, ib.cidx);
        if ((ib.flags & M355_IBF_HAS_RESIDUAL) && (stages & M355_STAGE_RESIDUAL)) {
          auto it = deferred.find(ib.res_ofs);
          if (it == deferred.end()) return -1;
          if (!run_rb(pic->rbs[it->second])) return -1;
        }
      }
    }
  }

  /* ---- in-loop filters ---- */
  if ((stages & M355_STAGE_DEBLOCK) && (pp.flags & M355_PF_DEBLOCK_ENABLED)) apply_deblocking_filter(img);
  if ((stages & M355_STAGE_SAO) && (pp.flags & M355_PF_SAO_ENABLED)) apply_sample_adaptive_offset_sequential(img);

  for (int c = 0; c < nc; c++) export_plane(img, c, out_planes[c]);
  (void)ilog2;
  return 0;
}


/* Pin for the SEI picture-hash path: builds a de265_image from tight planes and lets the reference's own
 * process_sei (sei.cc:441 -> process_sei_decoded_picture_hash, sei.cc:276-356) judge the candidate hash.
 * hash_type as sei.h:57-61.  Returns the reference's de265_error (0 = hash accepted, DE265_ERROR_CHECKSUM_MISMATCH
 * otherwise) or a negative code for setup errors. */
__attribute__((visibility("default")))
int m355_ref_check_hash(int width, int height, int chroma_format_idc, int bit_depth_luma, int bit_depth_chroma,
                        const void* const* planes, int hash_type, const uint8_t* md5 /* [3][16] */, const uint16_t* crc, const uint32_t* checksum)
{
  std::shared_ptr<seq_parameter_set> sps = std::make_shared<seq_parameter_set>();
  sps->set_defaults();
  sps->chroma_format_idc = chroma_format_idc;
  sps->pic_width_in_luma_samples = width; sps->pic_height_in_luma_samples = height;
  sps->bit_depth_luma = bit_depth_luma; sps->bit_depth_chroma = bit_depth_chroma;
  sps->log2_min_luma_coding_block_size = 3;
  sps->log2_diff_max_min_luma_coding_block_size = 0;
  sps->log2_min_transform_block_size = 2;
  sps->log2_diff_max_min_transform_block_size = 1;
  if (sps->compute_derived_values(false) != DE265_OK) return -2;
  sps->sps_read = true;
  std::shared_ptr<pic_parameter_set> pps = std::make_shared<pic_parameter_set>();
  pps->set_defaults();
  pps->sps = sps;
  pps->set_derived_values(sps.get());
  pps->pps_read = true;

  decoder_context dctx;
  dctx.param_sei_check_hash = true;              /* DE265_DECODER_PARAM_BOOL_SEI_CHECK_HASH (de265.cc:522); process_sei skips the check otherwise */
  de265_image img;
  if (img.alloc_image(width, height, (de265_chroma)chroma_format_idc, sps, false, &dctx, 0, nullptr, false) != DE265_OK) return -2;
  img.set_headers(nullptr, sps, pps);
  const int nc = chroma_format_idc ? 3 : 1;
  for (int c = 0; c < nc; c++) import_plane(&img, c, planes[c]);
  img.PicOutputFlag = true;

  sei_message sei;
  memset(&sei, 0, sizeof(sei));
  sei.payload_type = sei_payload_type_decoded_picture_hash;
  sei.data.decoded_picture_hash.hash_type = (sei_decoded_picture_hash_type)hash_type;
  if (md5) memcpy(sei.data.decoded_picture_hash.md5, md5, 48);
  if (crc) memcpy(sei.data.decoded_picture_hash.crc, crc, 6);
  if (checksum) memcpy(sei.data.decoded_picture_hash.checksum, checksum, 12);
  return (int)process_sei(&sei, &img);
}


/* CPU baseline harness (bench.py cpu_baseline): `threads` native threads, each replaying the picture into its own output planes
 * over and over for `seconds` (one picture stream per thread, like the reference's frame-parallel decoding) — no interpreter, no
 * shared buffers.  Returns the number of pictures replayed by all threads (negative: a replay failed); *elapsed = wall time. */
__attribute__((visibility("default")))
long m355_ref_bench(const m355_picture* pic, const void* const* ref_planes, int stages, int accel, int threads, double seconds, double* elapsed)
{
  const m355_pic_params& pp = pic->pp;
  const int sw = (pp.chroma_format_idc == 1 || pp.chroma_format_idc == 2) ? 2 : 1, sh = pp.chroma_format_idc == 1 ? 2 : 1;
  const size_t bps = pp.bit_depth_luma > 8 ? 2 : 1;
  const size_t nl = (size_t)pp.width * pp.height * bps, ncb = pp.chroma_format_idc ? (size_t)(pp.width / sw) * (pp.height / sh) * bps : 0;
  if (threads < 1) threads = 1;
  std::vector<long> counts((size_t)threads, 0);
  std::vector<int> rcs((size_t)threads, 0);
  const auto t0 = std::chrono::steady_clock::now();
  const auto stop = t0 + std::chrono::duration_cast<std::chrono::steady_clock::duration>(std::chrono::duration<double>(seconds));
  auto work = [&](int t) {
    std::vector<uint8_t> y(nl), cb(ncb ? ncb : 1), cr(ncb ? ncb : 1);
    void* out[3] = {y.data(), ncb ? cb.data() : nullptr, ncb ? cr.data() : nullptr};
    do {
      const int rc = m355_ref_replay(pic, ref_planes, stages, accel, out);
      if (rc) { rcs[(size_t)t] = rc; return; }
      counts[(size_t)t]++;
    } while (std::chrono::steady_clock::now() < stop);
  };
  std::vector<std::thread> th;
  for (int t = 1; t < threads; t++) th.emplace_back(work, t);
  work(0);
  for (auto& x : th) x.join();
  if (elapsed) *elapsed = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  long n = 0;
  for (int t = 0; t < threads; t++) { if (rcs[(size_t)t]) return rcs[(size_t)t]; n += counts[(size_t)t]; }
  return n;
}

} /* extern "C" */
