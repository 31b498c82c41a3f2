/*
 * ref_recorder.cc — TEST INFRASTRUCTURE (linked into oracle/_ref/libde265_ref.so only).
 *
 * Runs the REAL reference decoder (single-threaded, scalar table) on a bitstream and RECORDS, per
 * picture in decode order, the work lists of include/de265_mi355x.h — exactly what a libde265 built
 * with the glue patch of INTEGRATION.md would hand to m355_submit_picture() — together with the
 * reference's own final planes.  No reference source is modified: the recorder
 *   (a) overrides slots of decoder_context::acceleration (a public member, decctx.h:284) with
 *       wrappers that note the call and forward to the fallback, which captures coefficients,
 *       intra blocks (in decode order) and prediction-block rectangles/weights;
 *   (b) when a picture is complete, walks de265_image's metadata arrays (image.h:389-395) for
 *       CUs, transform-tree leaves, motion, QP, SAO and slice headers.
 * Used by tests/golden/make_girlshy_fixture.py to produce tests/golden/girlshy.m355rec.gz.
 *
 * Output file:  "M355REC1" | int32 n_pictures | per picture { int32 poc, dpb_idx, 0, 0 | blob }
 *               (blob = "M355WL01" work list, libde265_amd/worklist.py) ; the reference's planes
 *               of each picture (decode order, full uncropped planes, Y then Cb then Cr) go to
 *               <out>.planes ; display order + crop window go to <out>.order (text).
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "libde265/de265.h"
#include "libde265/decctx.h"
#include "libde265/image.h"
#include "libde265/slice.h"
#include "libde265/sps.h"
#include "libde265/pps.h"
#include "libde265/fallback.h"

#include "de265_mi355x.h"

namespace {

struct PicRec {
  de265_image* img = nullptr;
  int dpb_idx = -1;
  std::vector<m355_pb> pbs;
  std::vector<m355_wt> wts;
  std::vector<m355_rb> rbs;       // hook order; binned by size at finalize
  std::vector<m355_ib> ibs;
  std::vector<uint32_t> coeffs;
  uint32_t res_len = 0;
  int last_pb_luma = -1;          // index of the PB the following chroma calls belong to
  bool ref_usable[M355_MAX_REF_FRAMES] = {};  // DPB picture states while THIS picture decodes (motion.cc:365)
};

struct Pending {                  // coefficients seen by the dequant hook, waiting for their transform
  bool valid = false;
  std::vector<int16_t> lvl, pos;
  int32_t fact = 0, bdShift = 0;
};
struct PendingRes {               // skip/bypass residual waiting for add_residual to tell the position
  bool valid = false;
  int kind = 0, flags = 0, nT = 0;
  std::vector<int16_t> dense;
};

decoder_context* g_ctx = nullptr;
acceleration_functions g_orig;
PicRec g_cur;
Pending g_pend;
PendingRes g_pres;
bool g_rotated = false;
FILE* g_out = nullptr;
FILE* g_planes = nullptr;
int g_npics = 0;
std::string g_error;

struct Loc { de265_image* img; int dpb; int c, x, y; };

bool locate(const void* p, Loc* loc)
{
  for (int i = 0; g_ctx->has_image(i); i++) {
    de265_image* img = g_ctx->get_image(i);
    if (!img || !img->is_allocated()) continue;
    const int nc = img->get_chroma_format() == de265_chroma_mono ? 1 : 3;
    for (int c = 0; c < nc; c++) {
      const uint8_t* base = img->get_image_plane(c);
      const int bpp = img->get_bytes_per_pixel(c);
      const ptrdiff_t bytes = (ptrdiff_t)img->get_image_stride(c) * img->get_height(c) * bpp;
      const uint8_t* q = (const uint8_t*)p;
      if (q >= base && q < base + bytes) {
        const ptrdiff_t off = (q - base) / bpp;
        loc->img = img; loc->dpb = i; loc->c = c;
        loc->y = (int)(off / img->get_image_stride(c));
        loc->x = (int)(off % img->get_image_stride(c));
        return true;
      }
    }
  }
  return false;
}

void finalize_picture();

bool touch(const void* dst, Loc* loc)
{
  if (!locate(dst, loc)) { g_error = "slot called with a dst outside every DPB image"; return false; }
  if (loc->img != g_cur.img) {
    if (g_cur.img) finalize_picture();
    g_cur = PicRec();
    g_cur.img = loc->img;
    g_cur.dpb_idx = loc->dpb;
    for (int i = 0; i < M355_MAX_REF_FRAMES && g_ctx->has_image(i); i++) {
      const de265_image* rp = g_ctx->get_image(i);
      g_cur.ref_usable[i] = rp && rp->PicState != UnusedForReference;
    }
    g_pend.valid = false; g_pres.valid = false; g_rotated = false;
  }
  return true;
}

int log2i(int n) { int l = 0; while ((1 << l) < n) l++; return l; }

int qp_from_fact(int32_t fact)
{
  static const int ls[6] = {40, 45, 51, 57, 64, 72};
  for (int qp = 0; qp < 100; qp++)
    if ((int64_t)ls[qp % 6] << (qp / 6) == fact) return qp;
  return -1;
}

/* ------------------------------------------------------------------ residual bookkeeping ------ */

void push_coeffs(m355_rb& rb, const int16_t* lvl, const int16_t* pos, int n)
{
  rb.coeff_ofs = (uint32_t)g_cur.coeffs.size();
  rb.ncoeff = (uint16_t)n;
  for (int i = 0; i < n; i++) g_cur.coeffs.push_back((uint32_t)(uint16_t)pos[i] | ((uint32_t)(uint16_t)lvl[i] << 16));
}
void push_dense(m355_rb& rb, const int16_t* dense, int nT)
{
  rb.coeff_ofs = (uint32_t)g_cur.coeffs.size();
  int n = 0;
  for (int i = 0; i < nT * nT; i++)
    if (dense[i]) { g_cur.coeffs.push_back((uint32_t)i | ((uint32_t)(uint16_t)dense[i] << 16)); n++; }
  rb.ncoeff = (uint16_t)n;
}

/* called when the position of a residual block is known */
void commit_rb(m355_rb rb, const Loc& loc)
{
  rb.x = (uint16_t)loc.x; rb.y = (uint16_t)loc.y; rb.cidx = (uint8_t)loc.c;
  if (!g_cur.ibs.empty()) {
    m355_ib& ib = g_cur.ibs.back();
    if (ib.cidx == rb.cidx && ib.x == rb.x && ib.y == rb.y && ib.log2_size == rb.log2_size &&
        !(ib.flags & M355_IBF_HAS_RESIDUAL)) {
      ib.flags |= M355_IBF_HAS_RESIDUAL;
      ib.res_ofs = g_cur.res_len;
      rb.flags |= M355_RBF_DEFERRED;
      rb.res_ofs = g_cur.res_len;
      g_cur.res_len += 1u << (2 * rb.log2_size);
    }
  }
  g_cur.rbs.push_back(rb);
}

void on_transform(void* dst, const int16_t* coeffs, int log2nT, int is_dst)
{
  Loc loc;
  if (!touch(dst, &loc)) return;
  m355_rb rb; memset(&rb, 0, sizeof(rb));
  rb.log2_size = (uint8_t)log2nT;
  rb.kind = is_dst ? M355_RK_DST : M355_RK_DCT;
  const int bd = loc.img->get_bit_depth(loc.c);
  int qp = -1;
  if (g_pend.valid) {
    qp = qp_from_fact(g_pend.fact);
    if (g_pend.bdShift != bd + log2nT - 5 - 4) qp = -1;
  }
  if (qp >= 0) { rb.qp = (uint8_t)qp; push_coeffs(rb, g_pend.lvl.data(), g_pend.pos.data(), (int)g_pend.lvl.size()); }
  else { rb.flags |= M355_RBF_DEQUANTIZED; push_dense(rb, coeffs, 1 << log2nT); }
  g_pend.valid = false;
  commit_rb(rb, loc);
}

/* ------------------------------------------------------------------------------ slot hooks ---- */

#define T_ADD8(N, L) \
  void h_transform_add_8_##N(uint8_t* d, const int16_t* c, ptrdiff_t s) { on_transform(d, c, L, 0); g_orig.transform_add_8[L - 2](d, c, s); } \
  void h_transform_add_16_##N(uint16_t* d, const int16_t* c, ptrdiff_t s, int bd) { on_transform(d, c, L, 0); g_orig.transform_add_16[L - 2](d, c, s, bd); }
T_ADD8(4, 2) T_ADD8(8, 3) T_ADD8(16, 4) T_ADD8(32, 5)
void h_dst_add_8(uint8_t* d, const int16_t* c, ptrdiff_t s) { on_transform(d, c, 2, 1); g_orig.transform_4x4_dst_add_8(d, c, s); }
void h_dst_add_16(uint16_t* d, const int16_t* c, ptrdiff_t s, int bd) { on_transform(d, c, 2, 1); g_orig.transform_4x4_dst_add_16(d, c, s, bd); }

void h_dequant(int16_t* buf, const int16_t* lvl, const int16_t* pos, int n, int32_t fact, int32_t offset, int32_t bdShift)
{
  g_pend.valid = true;
  g_pend.lvl.assign(lvl, lvl + n); g_pend.pos.assign(pos, pos + n);
  g_pend.fact = fact; g_pend.bdShift = bdShift;
  g_orig.dequant_coeff_block(buf, lvl, pos, n, fact, offset, bdShift);
}
void h_rotate(int16_t* c, int nT) { g_rotated = true; g_orig.rotate_coefficients(c, nT); }

void note_res(int kind, int flags, const int16_t* coeffs, int nT)
{
  g_pres.valid = true; g_pres.kind = kind; g_pres.flags = flags; g_pres.nT = nT;
  g_pres.dense.assign(coeffs, coeffs + nT * nT);   /* post-dequant, post-rotation */
  g_pend.valid = false; g_rotated = false;
}
void h_skip_res(int32_t* r, const int16_t* c, int nT, int ts, int bs) { note_res(M355_RK_SKIP, 0, c, nT); g_orig.transform_skip_residual(r, c, nT, ts, bs); }
void h_rdpcm_v(int32_t* r, const int16_t* c, int nT, int ts, int bs) { note_res(M355_RK_SKIP, M355_RBF_RDPCM_V, c, nT); g_orig.rdpcm_v(r, c, nT, ts, bs); }
void h_rdpcm_h(int32_t* r, const int16_t* c, int nT, int ts, int bs) { note_res(M355_RK_SKIP, M355_RBF_RDPCM_H, c, nT); g_orig.rdpcm_h(r, c, nT, ts, bs); }
void h_bypass(int32_t* r, const int16_t* c, int nT) { note_res(M355_RK_BYPASS, 0, c, nT); g_orig.transform_bypass(r, c, nT); }
void h_bypass_v(int32_t* r, const int16_t* c, int nT) { note_res(M355_RK_BYPASS, M355_RBF_RDPCM_V, c, nT); g_orig.transform_bypass_rdpcm_v(r, c, nT); }
void h_bypass_h(int32_t* r, const int16_t* c, int nT) { note_res(M355_RK_BYPASS, M355_RBF_RDPCM_H, c, nT); g_orig.transform_bypass_rdpcm_h(r, c, nT); }
void on_add_residual(void* dst, int nT)
{
  Loc loc;
  if (!touch(dst, &loc)) return;
  if (!g_pres.valid || g_pres.nT != nT) { g_error = "add_residual without a recorded skip/bypass residual (cross-component prediction is not supported by the recorder)"; return; }
  m355_rb rb; memset(&rb, 0, sizeof(rb));
  rb.log2_size = (uint8_t)log2i(nT); rb.kind = (uint8_t)g_pres.kind;
  rb.flags = (uint8_t)(g_pres.flags | M355_RBF_DEQUANTIZED);
  push_dense(rb, g_pres.dense.data(), nT);
  g_pres.valid = false;
  commit_rb(rb, loc);
}
void h_add_res_8(uint8_t* d, ptrdiff_t s, const int32_t* r, int nT, int bd) { on_add_residual(d, nT); g_orig.add_residual_8(d, s, r, nT, bd); }
void h_add_res_16(uint16_t* d, ptrdiff_t s, const int32_t* r, int nT, int bd) { on_add_residual(d, nT); g_orig.add_residual_16(d, s, r, nT, bd); }

void on_intra(void* dst, int nT, int mode, int disable)
{
  Loc loc;
  if (!touch(dst, &loc)) return;
  m355_ib ib; memset(&ib, 0, sizeof(ib));
  ib.x = (uint16_t)loc.x; ib.y = (uint16_t)loc.y; ib.cidx = (uint8_t)loc.c;
  ib.log2_size = (uint8_t)log2i(nT); ib.mode = (uint8_t)mode;
  ib.flags = disable ? M355_IBF_DISABLE_BOUNDARY_FILTER : 0;
  g_cur.ibs.push_back(ib);
}
void h_dc_8(uint8_t* d, ptrdiff_t s, int nT, int c, const uint8_t* b) { on_intra(d, nT, 1, 0); g_orig.intra_pred_dc_8(d, s, nT, c, b); }
void h_dc_16(uint16_t* d, ptrdiff_t s, int nT, int c, const uint16_t* b) { on_intra(d, nT, 1, 0); g_orig.intra_pred_dc_16(d, s, nT, c, b); }
void h_pl_8(uint8_t* d, ptrdiff_t s, int nT, int c, const uint8_t* b) { on_intra(d, nT, 0, 0); g_orig.intra_pred_planar_8(d, s, nT, c, b); }
void h_pl_16(uint16_t* d, ptrdiff_t s, int nT, int c, const uint16_t* b) { on_intra(d, nT, 0, 0); g_orig.intra_pred_planar_16(d, s, nT, c, b); }
void h_an_8(uint8_t* d, ptrdiff_t s, int bd, int dis, int x0, int y0, int m, int nT, int c, const uint8_t* b) { on_intra(d, nT, m, dis); g_orig.intra_pred_angular_8(d, s, bd, dis, x0, y0, m, nT, c, b); }
void h_an_16(uint16_t* d, ptrdiff_t s, int bd, int dis, int x0, int y0, int m, int nT, int c, const uint16_t* b) { on_intra(d, nT, m, dis); g_orig.intra_pred_angular_16(d, s, bd, dis, x0, y0, m, nT, c, b); }

/* prediction write-back: kind 0 uni, 1 avg, 2 weighted uni, 3 weighted bi */
void on_pred(void* dst, int w, int h, int kind, int w1, int o1, int w2, int o2, int log2WD)
{
  Loc loc;
  if (!touch(dst, &loc)) return;
  if (loc.c == 0) {
    m355_pb pb; memset(&pb, 0, sizeof(pb));
    pb.x = (uint16_t)loc.x; pb.y = (uint16_t)loc.y; pb.w = (uint8_t)w; pb.h = (uint8_t)h;
    pb.reserved = (uint8_t)kind;   /* resolved at finalize */
    pb.ref_slot[0] = pb.ref_slot[1] = -1;
    if (kind >= 2) {
      pb.flags |= M355_PBF_WEIGHTED;
      m355_wt wt; memset(&wt, 0, sizeof(wt));
      wt.w[0] = (int16_t)w1; wt.o[0] = (int16_t)o1; wt.log2wd_luma = (uint8_t)log2WD;
      pb.wt_idx[0] = (uint16_t)g_cur.wts.size(); g_cur.wts.push_back(wt);
      wt.w[0] = (int16_t)w2; wt.o[0] = (int16_t)o2;
      pb.wt_idx[1] = (uint16_t)g_cur.wts.size(); g_cur.wts.push_back(wt);
    }
    g_cur.last_pb_luma = (int)g_cur.pbs.size();
    g_cur.pbs.push_back(pb);
  } else if (g_cur.last_pb_luma >= 0 && kind >= 2) {
    m355_pb& pb = g_cur.pbs[g_cur.last_pb_luma];
    m355_wt& a = g_cur.wts[pb.wt_idx[0]]; m355_wt& b = g_cur.wts[pb.wt_idx[1]];
    a.w[loc.c] = (int16_t)w1; a.o[loc.c] = (int16_t)o1; a.log2wd_chroma = (uint8_t)log2WD;
    b.w[loc.c] = (int16_t)w2; b.o[loc.c] = (int16_t)o2; b.log2wd_chroma = (uint8_t)log2WD;
  }
}
void h_un_8(uint8_t* d, ptrdiff_t ds, const int16_t* s, ptrdiff_t ss, int w, int h) { on_pred(d, w, h, 0, 0, 0, 0, 0, 0); g_orig.put_unweighted_pred_8(d, ds, s, ss, w, h); }
void h_un_16(uint16_t* d, ptrdiff_t ds, const int16_t* s, ptrdiff_t ss, int w, int h, int bd) { on_pred(d, w, h, 0, 0, 0, 0, 0, 0); g_orig.put_unweighted_pred_16(d, ds, s, ss, w, h, bd); }
void h_avg_8(uint8_t* d, ptrdiff_t ds, const int16_t* a, const int16_t* b, ptrdiff_t ss, int w, int h) { on_pred(d, w, h, 1, 0, 0, 0, 0, 0); g_orig.put_weighted_pred_avg_8(d, ds, a, b, ss, w, h); }
void h_avg_16(uint16_t* d, ptrdiff_t ds, const int16_t* a, const int16_t* b, ptrdiff_t ss, int w, int h, int bd) { on_pred(d, w, h, 1, 0, 0, 0, 0, 0); g_orig.put_weighted_pred_avg_16(d, ds, a, b, ss, w, h, bd); }
void h_wp_8(uint8_t* d, ptrdiff_t ds, const int16_t* s, ptrdiff_t ss, int w, int h, int wt, int o, int l) { on_pred(d, w, h, 2, wt, o, wt, o, l); g_orig.put_weighted_pred_8(d, ds, s, ss, w, h, wt, o, l); }
void h_wp_16(uint16_t* d, ptrdiff_t ds, const int16_t* s, ptrdiff_t ss, int w, int h, int wt, int o, int l, int bd) { on_pred(d, w, h, 2, wt, o, wt, o, l); g_orig.put_weighted_pred_16(d, ds, s, ss, w, h, wt, o, l, bd); }
void h_wb_8(uint8_t* d, ptrdiff_t ds, const int16_t* a, const int16_t* b, ptrdiff_t ss, int w, int h, int w1, int o1, int w2, int o2, int l) { on_pred(d, w, h, 3, w1, o1, w2, o2, l); g_orig.put_weighted_bipred_8(d, ds, a, b, ss, w, h, w1, o1, w2, o2, l); }
void h_wb_16(uint16_t* d, ptrdiff_t ds, const int16_t* a, const int16_t* b, ptrdiff_t ss, int w, int h, int w1, int o1, int w2, int o2, int l, int bd) { on_pred(d, w, h, 3, w1, o1, w2, o2, l); g_orig.put_weighted_bipred_16(d, ds, a, b, ss, w, h, w1, o1, w2, o2, l, bd); }

void install_hooks(acceleration_functions& a)
{
  g_orig = a;
  a.transform_add_8[0] = h_transform_add_8_4; a.transform_add_8[1] = h_transform_add_8_8;
  a.transform_add_8[2] = h_transform_add_8_16; a.transform_add_8[3] = h_transform_add_8_32;
  a.transform_add_16[0] = h_transform_add_16_4; a.transform_add_16[1] = h_transform_add_16_8;
  a.transform_add_16[2] = h_transform_add_16_16; a.transform_add_16[3] = h_transform_add_16_32;
  a.transform_4x4_dst_add_8 = h_dst_add_8; a.transform_4x4_dst_add_16 = h_dst_add_16;
  a.dequant_coeff_block = h_dequant; a.rotate_coefficients = h_rotate;
  a.transform_skip_residual = h_skip_res; a.rdpcm_v = h_rdpcm_v; a.rdpcm_h = h_rdpcm_h;
  a.transform_bypass = h_bypass; a.transform_bypass_rdpcm_v = h_bypass_v; a.transform_bypass_rdpcm_h = h_bypass_h;
  a.add_residual_8 = h_add_res_8; a.add_residual_16 = h_add_res_16;
  a.intra_pred_dc_8 = h_dc_8; a.intra_pred_dc_16 = h_dc_16; a.intra_pred_planar_8 = h_pl_8; a.intra_pred_planar_16 = h_pl_16;
  a.intra_pred_angular_8 = h_an_8; a.intra_pred_angular_16 = h_an_16;
  a.put_unweighted_pred_8 = h_un_8; a.put_unweighted_pred_16 = h_un_16;
  a.put_weighted_pred_avg_8 = h_avg_8; a.put_weighted_pred_avg_16 = h_avg_16;
  a.put_weighted_pred_8 = h_wp_8; a.put_weighted_pred_16 = h_wp_16;
  a.put_weighted_bipred_8 = h_wb_8; a.put_weighted_bipred_16 = h_wb_16;
}

/* ---------------------------------------------------------------- picture finalisation -------- */

void walk_tu(de265_image* img, int x0, int y0, int log2, int depth, std::vector<m355_tu>& out)
{
  /* same recursion as markTransformBlockBoundary (deblock.cc:33-63) */
  if (img->get_split_transform_flag(x0, y0, depth)) {
    const int h = 1 << (log2 - 1);
    walk_tu(img, x0, y0, log2 - 1, depth + 1, out);
    walk_tu(img, x0 + h, y0, log2 - 1, depth + 1, out);
    walk_tu(img, x0, y0 + h, log2 - 1, depth + 1, out);
    walk_tu(img, x0 + h, y0 + h, log2 - 1, depth + 1, out);
  } else {
    m355_tu tu; memset(&tu, 0, sizeof(tu));
    tu.x = (uint16_t)x0; tu.y = (uint16_t)y0; tu.log2_size = (uint8_t)log2;
    tu.flags = img->get_nonzero_coefficient(x0, y0) ? M355_TUF_NONZERO_COEFF : 0;
    out.push_back(tu);
  }
}

template <class T> void wr(const std::vector<T>& v)
{
  if (!v.empty()) fwrite(v.data(), sizeof(T), v.size(), g_out);
  size_t nb = v.size() * sizeof(T);
  static const char z[4] = {0, 0, 0, 0};
  if (nb % 4) fwrite(z, 1, 4 - nb % 4, g_out);
}

void finalize_picture()
{
  de265_image* img = g_cur.img;
  const seq_parameter_set& sps = img->get_sps();
  const pic_parameter_set& pps = img->get_pps();

  m355_pic_params pp; memset(&pp, 0, sizeof(pp));
  pp.width = sps.pic_width_in_luma_samples; pp.height = sps.pic_height_in_luma_samples;
  pp.chroma_format_idc = sps.chroma_format_idc;
  pp.bit_depth_luma = (uint8_t)sps.BitDepth_Y; pp.bit_depth_chroma = (uint8_t)sps.BitDepth_C;
  pp.log2_ctb_size = sps.Log2CtbSizeY; pp.log2_min_tb_size = (uint8_t)sps.Log2MinTrafoSize;
  pp.log2_min_cb_size = sps.Log2MinCbSizeY;
  pp.pic_cb_qp_offset = (int8_t)pps.pic_cb_qp_offset; pp.pic_cr_qp_offset = (int8_t)pps.pic_cr_qp_offset;
  if (pps.constrained_intra_pred_flag) pp.flags |= M355_PF_CONSTRAINED_INTRA_PRED;
  if (sps.strong_intra_smoothing_enable_flag) pp.flags |= M355_PF_STRONG_INTRA_SMOOTHING;
  if (sps.pcm_loop_filter_disable_flag) pp.flags |= M355_PF_PCM_LOOP_FILTER_DISABLE;
  if (pps.loop_filter_across_tiles_enabled_flag) pp.flags |= M355_PF_LF_ACROSS_TILES;
  if (sps.sample_adaptive_offset_enabled_flag && !g_ctx->param_disable_sao) pp.flags |= M355_PF_SAO_ENABLED;
  if (sps.range_extension.intra_smoothing_disabled_flag) pp.flags |= M355_PF_INTRA_SMOOTHING_DISABLED;
  if (sps.range_extension.implicit_rdpcm_enabled_flag) pp.flags |= M355_PF_IMPLICIT_RDPCM;
  if (sps.scaling_list_enable_flag) pp.flags |= M355_PF_SCALING_LIST;
  if (!g_ctx->param_disable_deblocking) pp.flags |= M355_PF_DEBLOCK_ENABLED;
  pp.num_tile_cols = pps.num_tile_columns; pp.num_tile_rows = pps.num_tile_rows;
  for (int i = 0; i <= pps.num_tile_columns; i++) pp.col_bd[i] = pps.colBd[i];
  for (int i = 0; i <= pps.num_tile_rows; i++) pp.row_bd[i] = pps.rowBd[i];

  std::vector<m355_slice> slices;
  for (size_t i = 0; i < img->slices.size(); i++) {
    const slice_segment_header* sh = img->slices[i];
    m355_slice s; memset(&s, 0, sizeof(s));
    s.slice_addr_rs = (int32_t)sh->SliceAddrRS;
    s.beta_offset = sh->slice_beta_offset; s.tc_offset = sh->slice_tc_offset;
    if (sh->slice_deblocking_filter_disabled_flag) s.flags |= M355_SF_DEBLOCK_DISABLED;
    if (sh->slice_loop_filter_across_slices_enabled_flag) s.flags |= M355_SF_LF_ACROSS_SLICES;
    if (sh->slice_sao_luma_flag) s.flags |= M355_SF_SAO_LUMA;
    if (sh->slice_sao_chroma_flag) s.flags |= M355_SF_SAO_CHROMA;
    slices.push_back(s);
  }

  const int ctbW = sps.PicWidthInCtbsY, ctbH = sps.PicHeightInCtbsY;
  std::vector<m355_ctb> ctbs(ctbW * ctbH);
  for (int y = 0; y < ctbH; y++)
    for (int x = 0; x < ctbW; x++) {
      m355_ctb& c = ctbs[y * ctbW + x]; memset(&c, 0, sizeof(c));
      c.slice_idx = img->get_SliceHeaderIndexCtb(x, y);
      const sao_info* sao = img->get_sao_info(x, y);
      c.sao_type = sao->SaoTypeIdx; c.sao_eo_class = sao->SaoEoClass;
      memcpy(c.sao_band_pos, sao->sao_band_position, 3);
      memcpy(c.sao_offset, sao->saoOffsetVal, 12);
      if (img->get_CTB_has_pcm_or_cu_transquant_bypass(x, y)) c.flags |= M355_CTBF_HAS_PCM_OR_BYPASS;
    }

  /* intra blocks are already in decode order; slice them per CTB */
  for (size_t i = 0; i < g_cur.ibs.size(); i++) {
    const m355_ib& ib = g_cur.ibs[i];
    const int sw = ib.cidx ? sps.SubWidthC : 1, sh = ib.cidx ? sps.SubHeightC : 1;
    const int ci = ((ib.y * sh) >> sps.Log2CtbSizeY) * ctbW + ((ib.x * sw) >> sps.Log2CtbSizeY);
    if (ctbs[ci].ib_count == 0) ctbs[ci].ib_start = (uint32_t)i;
    if (ctbs[ci].ib_start + ctbs[ci].ib_count != i) g_error = "intra blocks of one CTB are not contiguous in decode order";
    ctbs[ci].ib_count++;
  }

  std::vector<m355_cu> cus;
  std::vector<m355_tu> tus;
  const int minCb = sps.MinCbSizeY;
  for (int cy = 0; cy < sps.PicHeightInMinCbsY; cy++)
    for (int cx = 0; cx < sps.PicWidthInMinCbsY; cx++) {
      const int l2 = img->get_log2CbSize_cbUnits(cx, cy);
      if (l2 == 0) continue;
      const int x0 = cx * minCb, y0 = cy * minCb;
      m355_cu cu; memset(&cu, 0, sizeof(cu));
      cu.x = (uint16_t)x0; cu.y = (uint16_t)y0; cu.log2_size = (uint8_t)l2;
      cu.pred_mode = (uint8_t)img->get_pred_mode(x0, y0);
      cu.part_mode = (uint8_t)img->get_PartMode(x0, y0);
      cu.qp_y = (int8_t)img->get_QPY(x0, y0);
      if (img->get_pcm_flag(x0, y0)) { cu.flags |= M355_CUF_PCM; g_error = "PCM coding units are not supported by the recorder"; }
      if (img->get_cu_transquant_bypass(x0, y0)) cu.flags |= M355_CUF_TRANSQUANT_BYPASS;
      cus.push_back(cu);
      walk_tu(img, x0, y0, l2, 0, tus);
    }

  /* prediction blocks: motion from pb_info, reference identity = DPB index (motion.cc:352-353) */
  int32_t ref_frames[M355_MAX_REF_FRAMES];
  for (int i = 0; i < M355_MAX_REF_FRAMES; i++) ref_frames[i] = -1;
  for (size_t i = 0; i < g_cur.pbs.size(); i++) {
    m355_pb& pb = g_cur.pbs[i];
    const int kind = pb.reserved; pb.reserved = 0;
    const PBMotion& mi = img->get_mv_info(pb.x, pb.y);
    const slice_segment_header* sh = img->get_SliceHeader(pb.x, pb.y);
    for (int l = 0; l < 2; l++) {
      if (!mi.predFlag[l]) continue;
      pb.flags |= (M355_PBF_PRED_L0 << l);
      const int dpb = sh->RefPicList[l][mi.refIdx[l]];
      if (dpb < 0 || dpb >= M355_MAX_REF_FRAMES) { g_error = "reference index outside the recordable DPB range"; continue; }
      pb.ref_slot[l] = (int8_t)dpb;
      ref_frames[dpb] = dpb;
      pb.mv[l][0] = mi.mv[l].x; pb.mv[l][1] = mi.mv[l].y;
      if (!g_cur.ref_usable[dpb]) pb.flags |= (M355_PBF_FILL_L0 << l);
    }
    const bool bi = (kind == 1 || kind == 3);
    if (bi) pb.flags |= M355_PBF_MC_L0 | M355_PBF_MC_L1;
    else pb.flags |= mi.predFlag[0] ? M355_PBF_MC_L0 : M355_PBF_MC_L1;   /* incl. the identical-MV demotion (motion.cc:348-357) */
    if (kind == 2 && !mi.predFlag[0]) {    /* uni-weighted from list 1: its weights were stored in slot 0 */
      std::swap(pb.wt_idx[0], pb.wt_idx[1]);
    }
  }

  /* residual blocks: bin by size (stable) */
  std::vector<m355_rb> rbs; int32_t rb_count[4] = {0, 0, 0, 0};
  for (int s = 2; s <= 5; s++)
    for (size_t i = 0; i < g_cur.rbs.size(); i++)
      if (g_cur.rbs[i].log2_size == s) { rbs.push_back(g_cur.rbs[i]); rb_count[s - 2]++; }

  /* ---- write ---- */
  int32_t hdr[4] = {img->PicOrderCntVal, g_cur.dpb_idx, 0, 0};
  fwrite(hdr, 4, 4, g_out);
  fwrite("M355WL01", 1, 8, g_out);
  fwrite(&pp, sizeof(pp), 1, g_out);
  int32_t dst = g_cur.dpb_idx; fwrite(&dst, 4, 1, g_out);
  fwrite(ref_frames, 4, M355_MAX_REF_FRAMES, g_out);
  int32_t counts[15] = {(int32_t)slices.size(), (int32_t)ctbs.size(), (int32_t)cus.size(), (int32_t)tus.size(),
                        (int32_t)g_cur.pbs.size(), (int32_t)g_cur.wts.size(), (int32_t)g_cur.ibs.size(),
                        rb_count[0], rb_count[1], rb_count[2], rb_count[3], (int32_t)g_cur.coeffs.size(), 0,
                        (int32_t)g_cur.res_len, 0};
  fwrite(counts, 4, 15, g_out);
  wr(slices); wr(ctbs); wr(cus); wr(tus); wr(g_cur.pbs); wr(g_cur.wts); wr(rbs); wr(g_cur.ibs); wr(g_cur.coeffs);
  /* pcm: none */

  /* reference planes (full, uncropped) */
  const int nc = img->get_chroma_format() == de265_chroma_mono ? 1 : 3;
  for (int c = 0; c < nc; c++) {
    const int bpp = img->get_bytes_per_pixel(c);
    for (int y = 0; y < img->get_height(c); y++)
      fwrite(img->get_image_plane(c) + (size_t)y * img->get_image_stride(c) * bpp, bpp, img->get_width(c), g_planes);
  }
  g_npics++;
}

} // namespace

extern "C" int ref_record_stream(const char* h265_path, const char* out_path, int disable_deblocking, int disable_sao)
{
  g_error.clear(); g_npics = 0; g_cur = PicRec();
  FILE* in = fopen(h265_path, "rb");
  if (!in) return -1;
  g_out = fopen(out_path, "wb");
  g_planes = fopen((std::string(out_path) + ".planes").c_str(), "wb");
  FILE* order = fopen((std::string(out_path) + ".order").c_str(), "w");
  if (!g_out || !g_planes || !order) return -2;
  fwrite("M355REC1", 1, 8, g_out);
  int32_t zero = 0; fwrite(&zero, 4, 1, g_out);

  de265_decoder_context* ctx = de265_new_decoder();
  de265_set_parameter_int(ctx, DE265_DECODER_PARAM_ACCELERATION_CODE, de265_acceleration_SCALAR);
  de265_set_parameter_bool(ctx, DE265_DECODER_PARAM_DISABLE_DEBLOCKING, disable_deblocking);
  de265_set_parameter_bool(ctx, DE265_DECODER_PARAM_DISABLE_SAO, disable_sao);
  g_ctx = (decoder_context*)ctx;
  install_hooks(g_ctx->acceleration);

  /* push the whole stream, then decode NAL by NAL (same loop shape as dec265/dec265.cc:790-838) */
  {
    std::vector<uint8_t> data;
    uint8_t buf[65536];
    size_t n;
    while ((n = fread(buf, 1, sizeof(buf), in)) > 0) data.insert(data.end(), buf, buf + n);
    de265_push_data(ctx, data.data(), (int)data.size(), 0, nullptr);
    de265_flush_data(ctx);
  }
  int more = 1;
  while (more) {
    more = 0;
    de265_error err = de265_decode(ctx, &more);
    if (err != DE265_OK) break;
    const de265_image* o;
    while ((o = de265_get_next_picture(ctx)) != nullptr) {
      int stride;
      const uint8_t* p0 = de265_get_image_plane(o, 0, &stride);
      const de265_image* im = (const de265_image*)o;
      const ptrdiff_t d = (p0 - im->get_image_plane(0)) / im->get_bytes_per_pixel(0);
      fprintf(order, "%d %d %d %d %d\n", im->PicOrderCntVal, de265_get_image_width(o, 0), de265_get_image_height(o, 0),
              (int)(d % im->get_image_stride(0)), (int)(d / im->get_image_stride(0)));
    }
  }
  if (g_cur.img) finalize_picture();
  g_cur = PicRec();
  de265_free_decoder(ctx);
  g_ctx = nullptr;
  fseek(g_out, 8, SEEK_SET);
  int32_t n = g_npics; fwrite(&n, 4, 1, g_out);
  fclose(g_out); fclose(g_planes); fclose(order); fclose(in);
  if (!g_error.empty()) { fprintf(stderr, "ref_record_stream: %s\n", g_error.c_str()); return -3; }
  return g_npics;
}
