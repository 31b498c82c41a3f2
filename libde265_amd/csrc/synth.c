/*
 * synth.c — deterministic synthetic work-list generator (host, plain C) for the benchmark configs of
 * BASELINE.json / SURVEY.md §8(d): there are no 4K/8K bitstreams (and no working encoder) offline,
 * so workloads are generated at the work-list level — the level the pixel path consumes.
 *
 * Recipe: xorshift32 (the generator of the reference's dev-tools/test-*.cc); per CTB a
 * uniform choice of CU size 64/32/16/8; intra CUs with uniform modes 0..34 (luma 4x4 -> DST) or inter
 * CUs with 2Nx2N / 2NxN / Nx2N / AMP partitions, MVs uniform in +-64 quarter-pels over all fractional
 * phases, 50 % bi-prediction, 10 % explicit weights (parser ranges, slice.cc:159-231), 2 % MVs far
 * outside the picture; coefficient blocks in the three scenarios of dev-tools/test-transform.cc:61-73
 * (sparse 1..N/8 in +-512, dense +-2048, full int16) mixed 80/15/5 %; QP uniform 22..37; SAO type per
 * CTB uniform {off, band, edge x 4}; uniform tiles.  Every list is structurally valid HEVC decode
 * order (z-scan inside a CTB, tile scan across CTBs) so the intra wavefront sees real dependencies.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "de265_mi355x.h"

typedef struct m355_synth_cfg {
  int32_t width, height;
  int32_t bit_depth;        /* luma = chroma */
  int32_t log2_ctb;         /* 4..6 */
  int32_t tile_cols, tile_rows;
  int32_t intra_pct;        /* 100 = intra picture (config C2), else % of CUs coded intra */
  int32_t bipred_pct, weighted_pct, oob_mv_pct;
  int32_t cbf_pct;          /* probability that a transform block carries coefficients */
  int32_t deblock, sao;
  int32_t n_refs;           /* reference frames available (slots 0..n_refs-1) */
  int32_t lf_across_tiles;
  uint32_t seed;
  int32_t fixed_cu_log2;    /* 0 = random per CTB, else force this CU size (3..6) */
  int32_t n_slices;         /* 0/1 = one slice; else that many slices with random filter flags / offsets (whole tiles per
                               slice when the picture has tiles, arbitrary CTB runs otherwise) */
  int32_t features;         /* M355_SYN_* bits */
  int32_t chroma_format;    /* 0 or 1 = 4:2:0 (default), 2 = 4:2:2, 3 = 4:4:4, 4 = monochrome */
} m355_synth_cfg;
enum { M355_SYN_CONSTRAINED_INTRA = 1, M355_SYN_TRANSQUANT_BYPASS = 2, M355_SYN_SCALING_LIST = 4, M355_SYN_PCM = 8,
       M355_SYN_PCM_LOOP_FILTER_DISABLE = 16, M355_SYN_CROSS_COMPONENT = 32 /* 4:4:4 only */,
       M355_SYN_RDPCM = 64,          /* RExt: transform skip up to 32x32, implicit (intra, modes 10 / 26) and explicit (inter) RDPCM on
                                        skip and bypass blocks (slice.cc:3491-3503, fallback-dct.cc:161-256) */
       M355_SYN_ROTATE = 128,        /* RExt transform_skip_rotation_enabled_flag: 4x4 skip / bypass blocks of intra CUs (transform.cc:400-402) */
       M355_SYN_MISSING_REF = 256,   /* some PBs reference a DPB slot that holds no picture: predSamples = 1 << 13 (motion.cc:362-376) */
       M355_SYN_DEQUANTIZED = 512    /* some blocks carry already-scaled levels (what the table slots receive) */ };

typedef struct { void* p; size_t n, cap, esz; } vec;
typedef struct gen {
  const m355_synth_cfg* cfg;
  uint32_t s;
  vec slices, ctbs, cus, tus, pbs, wts, rbs[4], ibs, coeffs, pcm;
  uint32_t res_len;
  int ctbW, ctbH;
  int cf;                   /* chroma_format_idc of the picture */
  int ctb_has_bypass;       /* a CU of the CTB being generated uses cu_transquant_bypass */
  int cu_bypass;            /* the CU being generated */
} gen;

static uint32_t rnd(gen* g) { uint32_t s = g->s; s ^= s << 13; s ^= s >> 17; s ^= s << 5; g->s = s; return s; }
static int rbelow(gen* g, int n) { return (int)(rnd(g) % (uint32_t)n); }
static int rrange(gen* g, int lo, int hi) { return lo + (int)(rnd(g) % (uint32_t)(hi - lo + 1)); }
static int pct(gen* g, int p) { return rbelow(g, 100) < p; }

static void* vpush(vec* v, size_t esz)
{
  v->esz = esz;
  if (v->n == v->cap) { v->cap = v->cap ? v->cap * 2 : 1024; v->p = realloc(v->p, v->cap * esz); }
  void* e = (char*)v->p + v->n * esz;
  memset(e, 0, esz);
  v->n++;
  return e;
}

/* coefficient block in one of the three scenarios of dev-tools/test-transform.cc:61-73 */
static void gen_coeffs(gen* g, m355_rb* rb, int n)
{
  rb->coeff_ofs = (uint32_t)g->coeffs.n;
  const int r = rbelow(g, 100);
  int cnt = 0;
  if (r < 80) { /* sparse, low-frequency biased like a real residual */
    const int nnz = 1 + rbelow(g, n / 8 > 0 ? n / 8 : 1);
    int nT = 1; while (nT * nT < n) nT <<= 1;
    uint8_t used[1024]; memset(used, 0, (size_t)n);
    for (int i = 0; i < nnz; i++) {
      const int lim = (rbelow(g, 4) == 0) ? nT : (nT > 4 ? nT / 2 : nT);
      const int pos = rbelow(g, lim) + rbelow(g, lim) * nT;
      if (used[pos]) continue;
      used[pos] = 1;
      int v = rrange(g, -512, 512); if (v == 0) v = 1;
      *(uint32_t*)vpush(&g->coeffs, 4) = (uint32_t)pos | ((uint32_t)(uint16_t)(int16_t)v << 16);
      cnt++;
    }
  } else {
    const int lo = r < 95 ? -2048 : -32768, hi = r < 95 ? 2048 : 32767;
    for (int pos = 0; pos < n; pos++) {
      int v = rrange(g, lo, hi); if (v == 0) continue;
      *(uint32_t*)vpush(&g->coeffs, 4) = (uint32_t)pos | ((uint32_t)(uint16_t)(int16_t)v << 16);
      cnt++;
    }
  }
  rb->ncoeff = (uint16_t)cnt;
}

/* one transform block of one component; returns 1 if it carries coefficients */
static int gen_tb(gen* g, int cidx, int x, int y, int log2, int intra, int mode, int qp, uint32_t* first_ib)
{
  const int n = 1 << (2 * log2);
  const int cbf = pct(g, g->cfg->cbf_pct);
  m355_ib* ib = NULL;
  (void)first_ib;
  if (intra) {
    ib = (m355_ib*)vpush(&g->ibs, sizeof(m355_ib));
    ib->x = (uint16_t)x; ib->y = (uint16_t)y; ib->cidx = (uint8_t)cidx; ib->log2_size = (uint8_t)log2; ib->mode = (uint8_t)mode;
  }
  if (!cbf) return 0;
  m355_rb* rb = (m355_rb*)vpush(&g->rbs[log2 - 2], sizeof(m355_rb));
  rb->x = (uint16_t)x; rb->y = (uint16_t)y; rb->cidx = (uint8_t)cidx; rb->log2_size = (uint8_t)log2;
  rb->qp = (uint8_t)(cidx ? (qp > 3 ? qp - 3 : qp) : qp);
  rb->kind = (intra && cidx == 0 && log2 == 2) ? M355_RK_DST : M355_RK_DCT;
  if (log2 == 2 && rbelow(g, 20) == 0) rb->kind = M355_RK_SKIP;   /* transform_skip is Main profile for 4x4 */
  if ((g->cfg->features & M355_SYN_RDPCM) && rbelow(g, 4) == 0) rb->kind = M355_RK_SKIP;   /* log2_max_transform_skip_block_size up to 5 (RExt) */
  if (g->cfg->features & M355_SYN_SCALING_LIST) {                  /* matrixID rule of transform.cc:493-502 */
    int m = log2 == 5 ? 0 : cidx;
    if (!intra) m += (log2 < 5) ? 3 : 1;
    rb->matrix_id = (uint8_t)m;
  }
  if (g->cu_bypass) {                                              /* cu_transquant_bypass: the levels ARE the residual */
    rb->kind = M355_RK_BYPASS;
    rb->coeff_ofs = (uint32_t)g->coeffs.n;
    int cnt = 0;
    for (int pos = 0; pos < n; pos++) {
      if (rbelow(g, 3)) continue;
      int v = rrange(g, -40, 40); if (v == 0) v = 1;
      *(uint32_t*)vpush(&g->coeffs, 4) = (uint32_t)pos | ((uint32_t)(uint16_t)(int16_t)v << 16);
      cnt++;
    }
    rb->ncoeff = (uint16_t)cnt;
  } else
  gen_coeffs(g, rb, n);
  if ((g->cfg->features & M355_SYN_RDPCM) && (rb->kind == M355_RK_SKIP || rb->kind == M355_RK_BYPASS)) {
    /* intra: implicit_rdpcm_enabled_flag && mode 10 (horizontal) / 26 (vertical); inter: explicit_rdpcm_flag + dir (slice.cc:3491-3503) */
    if (intra) rb->flags |= (uint8_t)(mode == 10 ? M355_RBF_RDPCM_H : (mode == 26 ? M355_RBF_RDPCM_V : 0));
    else { const int r3 = rbelow(g, 3); rb->flags |= (uint8_t)(r3 == 1 ? M355_RBF_RDPCM_H : (r3 == 2 ? M355_RBF_RDPCM_V : 0)); }
  }
  if ((g->cfg->features & M355_SYN_DEQUANTIZED) && rb->kind != M355_RK_BYPASS && rbelow(g, 3) == 0) rb->flags |= M355_RBF_DEQUANTIZED;
  if (intra) {
    rb->flags |= M355_RBF_DEFERRED; rb->res_ofs = g->res_len;
    ib->flags |= M355_IBF_HAS_RESIDUAL; ib->res_ofs = g->res_len;
    g->res_len += (uint32_t)n;
  }
  return 1;
}

/* transform unit at luma (x,y,log2): luma block, then chroma blocks (4:2:0), decode order
 * (slice.cc:3706-3847): chroma of four 4x4 luma blocks is one 4x4 pair emitted after the 4th. */
static void gen_tu(gen* g, int x, int y, int log2, int intra, int lmode, int cmode, int qp, int blk_idx, int xbase, int ybase)
{
  const int nz = gen_tb(g, 0, x, y, log2, intra, lmode, qp, NULL);
  m355_tu* tu = (m355_tu*)vpush(&g->tus, sizeof(m355_tu));
  tu->x = (uint16_t)x; tu->y = (uint16_t)y; tu->log2_size = (uint8_t)log2; tu->flags = nz ? M355_TUF_NONZERO_COEFF : 0;
  /* chroma blocks of the transform unit (slice.cc:3706-3847): 4:2:0 halves both dimensions (the chroma of four 4x4
     luma blocks is one 4x4 pair emitted after the 4th); 4:2:2 halves the width only — two square blocks stacked
     per component; 4:4:4 follows luma down to 4x4 */
  if (g->cf == 1) {
    if (log2 > 2) {
      gen_tb(g, 1, x / 2, y / 2, log2 - 1, intra, cmode, qp, NULL);
      gen_tb(g, 2, x / 2, y / 2, log2 - 1, intra, cmode, qp, NULL);
    } else if (blk_idx == 3) {
      gen_tb(g, 1, xbase / 2, ybase / 2, 2, intra, cmode, qp, NULL);
      gen_tb(g, 2, xbase / 2, ybase / 2, 2, intra, cmode, qp, NULL);
    }
  } else if (g->cf == 2) {
    for (int c = 1; c <= 2; c++) {
      if (log2 > 2) {
        gen_tb(g, c, x / 2, y, log2 - 1, intra, cmode, qp, NULL);
        gen_tb(g, c, x / 2, y + (1 << (log2 - 1)), log2 - 1, intra, cmode, qp, NULL);
      } else if (blk_idx == 3) {
        gen_tb(g, c, xbase / 2, ybase, 2, intra, cmode, qp, NULL);
        gen_tb(g, c, xbase / 2, ybase + 4, 2, intra, cmode, qp, NULL);
      }
    }
  } else if (g->cf == 3) {
    /* cross-component prediction (slice.cc:3721-3760): only with cbf_luma; a chroma block with cbf 0 but a
       ResScaleVal is still listed (ncoeff 0); matrix_id bits 3..7 carry distance-to-luma / magnitude / sign */
    const int ccp_ok = (g->cfg->features & M355_SYN_CROSS_COMPONENT) && nz;
    int emitted = 0;
    for (int c = 1; c <= 2; c++) {
      const int v = (ccp_ok && rbelow(g, 2)) ? 1 + rbelow(g, 4) : 0, neg = v ? rbelow(g, 2) : 0;
      const size_t before = g->rbs[log2 - 2].n;
      const int had = gen_tb(g, c, x, y, log2, intra, cmode, qp, NULL);
      if (!had && v) {
        /* gen_tb pushed no block (cbf 0): list an empty one.  For intra blocks gen_tb has pushed the ib already. */
        m355_rb* rb = (m355_rb*)vpush(&g->rbs[log2 - 2], sizeof(m355_rb));
        rb->x = (uint16_t)x; rb->y = (uint16_t)y; rb->cidx = (uint8_t)c; rb->log2_size = (uint8_t)log2;
        rb->qp = (uint8_t)(qp > 3 ? qp - 3 : qp); rb->kind = g->cu_bypass ? M355_RK_BYPASS : M355_RK_DCT;
        rb->coeff_ofs = (uint32_t)g->coeffs.n; rb->ncoeff = 0;
        if (g->cfg->features & M355_SYN_SCALING_LIST) rb->matrix_id = (uint8_t)((log2 == 5 ? 0 : c) + (intra ? 0 : (log2 < 5 ? 3 : 1)));
        if (intra) {
          m355_ib* ib = (m355_ib*)g->ibs.p + (g->ibs.n - 1);
          const uint32_t n2 = 1u << (2 * log2);
          rb->flags |= M355_RBF_DEFERRED; rb->res_ofs = g->res_len;
          ib->flags |= M355_IBF_HAS_RESIDUAL; ib->res_ofs = g->res_len;
          g->res_len += n2;
        }
      }
      if (g->rbs[log2 - 2].n > before) {
        m355_rb* rb = (m355_rb*)g->rbs[log2 - 2].p + (g->rbs[log2 - 2].n - 1);
        if (v) rb->matrix_id |= (uint8_t)((v << 4) | (neg << 7) | (emitted ? 8 : 0));
        emitted++;
      }
    }
  }
}

static void gen_tt(gen* g, int x, int y, int log2, int depth, int intra, const int* lmodes, int cmode, int qp, int nxn)
{
  /* forced split above 32, intra NxN splits once, otherwise split once with probability 1/3 */
  int split = log2 > 5 || (nxn && depth == 0) || (log2 > 2 && depth == 0 && !nxn && rbelow(g, 3) == 0);
  if (split) {
    const int h = 1 << (log2 - 1);
    for (int i = 0; i < 4; i++) {
      const int cx = x + (i & 1) * h, cy = y + (i >> 1) * h;
      if (log2 - 1 == 2) gen_tu(g, cx, cy, 2, intra, lmodes[nxn ? i : 0], cmode, qp, i, x, y);
      else gen_tt(g, cx, cy, log2 - 1, depth + 1, intra, lmodes + 0, cmode, qp, 0);
    }
  } else gen_tu(g, x, y, log2, intra, lmodes[0], cmode, qp, 0, x, y);
}

static void gen_pb(gen* g, int x, int y, int w, int h)
{
  const m355_synth_cfg* c = g->cfg;
  m355_pb* pb = (m355_pb*)vpush(&g->pbs, sizeof(m355_pb));
  pb->x = (uint16_t)x; pb->y = (uint16_t)y; pb->w = (uint8_t)w; pb->h = (uint8_t)h;
  const int small = (w + h) <= 12; /* 8x4 / 4x8: uni-prediction only */
  const int bi = !small && c->n_refs >= 1 && pct(g, c->bipred_pct);
  const int l_only = bi ? 0 : rbelow(g, 2);
  pb->ref_slot[0] = pb->ref_slot[1] = -1;
  for (int l = 0; l < 2; l++) {
    if (!bi && l != l_only) continue;
    pb->flags |= (uint8_t)((M355_PBF_PRED_L0 | M355_PBF_MC_L0) << l);
    pb->ref_slot[l] = (int8_t)rbelow(g, c->n_refs);
    int mvx = rrange(g, -64, 64), mvy = rrange(g, -64, 64);
    if (pct(g, c->oob_mv_pct)) { mvx = rrange(g, -4 * c->width, 4 * c->width); mvy = rrange(g, -4 * c->height, 4 * c->height); if (mvx > 32000) mvx = 32000; if (mvx < -32000) mvx = -32000; if (mvy > 32000) mvy = 32000; if (mvy < -32000) mvy = -32000; }
    pb->mv[l][0] = (int16_t)mvx; pb->mv[l][1] = (int16_t)mvy;
    if ((c->features & M355_SYN_MISSING_REF) && rbelow(g, 8) == 0) {   /* RefPicList entry without a usable picture */
      pb->flags |= (uint8_t)(M355_PBF_FILL_L0 << l);
      pb->ref_slot[l] = (int8_t)c->n_refs;
    }
  }
  if (pct(g, c->weighted_pct)) {
    pb->flags |= M355_PBF_WEIGHTED;
    const int denom_l = rbelow(g, 8), denom_c = rbelow(g, 8);
    const int shift1 = 14 - c->bit_depth < 2 ? 2 : 14 - c->bit_depth;
    for (int l = 0; l < 2; l++) {
      m355_wt* wt = (m355_wt*)vpush(&g->wts, sizeof(m355_wt));
      pb->wt_idx[l] = (uint16_t)((g->wts.n - 1) & 0xFFFF);
      for (int k = 0; k < 3; k++) {
        wt->w[k] = (int16_t)((1 << (k ? denom_c : denom_l)) + rrange(g, -128, 127));
        wt->o[k] = (int16_t)(rrange(g, -128, 127) * (1 << (c->bit_depth - 8)));
      }
      wt->log2wd_luma = (uint8_t)(denom_l + shift1); wt->log2wd_chroma = (uint8_t)(denom_c + shift1);
    }
    if (g->wts.n > 65000) { pb->flags &= (uint8_t)~M355_PBF_WEIGHTED; g->wts.n -= 2; }
  }
}

static void gen_cu(gen* g, int x, int y, int log2)
{
  const m355_synth_cfg* c = g->cfg;
  m355_cu* cu = (m355_cu*)vpush(&g->cus, sizeof(m355_cu));
  const int intra = c->n_refs == 0 || pct(g, c->intra_pct);
  const int qp = rrange(g, 22, 37);
  const int size = 1 << log2;
  cu->x = (uint16_t)x; cu->y = (uint16_t)y; cu->log2_size = (uint8_t)log2; cu->qp_y = (int8_t)qp;
  g->cu_bypass = (c->features & M355_SYN_TRANSQUANT_BYPASS) && rbelow(g, 12) == 0;
  if (g->cu_bypass) { cu->flags |= M355_CUF_TRANSQUANT_BYPASS; g->ctb_has_bypass = 1; }
  if (intra && !g->cu_bypass && log2 <= 5 && (c->features & M355_SYN_PCM) && rbelow(g, 10) == 0) {
    /* PCM coding unit (slice.cc:4211-4255): raw samples, no prediction, no transform tree */
    cu->pred_mode = 0; cu->part_mode = 0; cu->flags |= M355_CUF_PCM; g->ctb_has_bypass = 1;
    const int ncomp = g->cf ? 3 : 1;
    for (int cidx = 0; cidx < ncomp; cidx++) {
      /* square raw blocks: 4:2:2 chroma (half width, full height) is two stacked squares */
      const int subw = (cidx && (g->cf == 1 || g->cf == 2)) ? 1 : 0, subh = (cidx && g->cf == 1) ? 1 : 0;
      const int l2 = log2 - subw, nblk = (subw && !subh) ? 2 : 1, n = 1 << (2 * l2);
      const int pcm_bits = c->bit_depth - rbelow(g, 3);           /* PcmBitDepth <= BitDepth: samples << (BitDepth - PcmBitDepth) */
      for (int b = 0; b < nblk; b++) {
        m355_ib* ib = (m355_ib*)vpush(&g->ibs, sizeof(m355_ib));
        ib->x = (uint16_t)(x >> subw); ib->y = (uint16_t)((y >> subh) + b * (1 << l2)); ib->cidx = (uint8_t)cidx; ib->log2_size = (uint8_t)l2;
        ib->mode = 1; ib->flags = M355_IBF_PCM; ib->res_ofs = (uint32_t)g->pcm.n;
        for (int i = 0; i < n; i++) *(uint16_t*)vpush(&g->pcm, 2) = (uint16_t)(rbelow(g, 1 << pcm_bits) << (c->bit_depth - pcm_bits));
      }
    }
    m355_tu* tu = (m355_tu*)vpush(&g->tus, sizeof(m355_tu));
    tu->x = (uint16_t)x; tu->y = (uint16_t)y; tu->log2_size = (uint8_t)log2;
  } else if (intra) {
    cu->pred_mode = 0;
    const int nxn = (log2 == 3) && rbelow(g, 2);
    cu->part_mode = nxn ? 3 : 0;
    int lmodes[4];
    for (int i = 0; i < 4; i++) lmodes[i] = rbelow(g, 35);
    int cmode = rbelow(g, 35);
    if (c->features & M355_SYN_RDPCM) {                                /* implicit RDPCM needs modes 10 / 26: make them common */
      for (int i = 0; i < 4; i++) if (rbelow(g, 2)) lmodes[i] = rbelow(g, 2) ? 10 : 26;
      if (rbelow(g, 2)) cmode = rbelow(g, 2) ? 10 : 26;
    }
    gen_tt(g, x, y, log2, 0, 1, lmodes, cmode, qp, nxn);
  } else {
    const int skip = rbelow(g, 4) == 0;
    cu->pred_mode = skip ? 2 : 1;
    int pm = 0;
    if (!skip) {
      const int r = rbelow(g, 10);
      if (r < 5) pm = 0; else if (r < 7) pm = 1; else if (r < 9) pm = 2;
      else pm = (log2 >= 4) ? 4 + rbelow(g, 4) : 0;
    }
    cu->part_mode = (uint8_t)pm;
    const int h2 = size / 2, q4 = size / 4;
    switch (pm) {
      case 0: gen_pb(g, x, y, size, size); break;
      case 1: gen_pb(g, x, y, size, h2); gen_pb(g, x, y + h2, size, h2); break;
      case 2: gen_pb(g, x, y, h2, size); gen_pb(g, x + h2, y, h2, size); break;
      case 4: gen_pb(g, x, y, size, q4); gen_pb(g, x, y + q4, size, size - q4); break;
      case 5: gen_pb(g, x, y, size, size - q4); gen_pb(g, x, y + size - q4, size, q4); break;
      case 6: gen_pb(g, x, y, q4, size); gen_pb(g, x + q4, y, size - q4, size); break;
      default: gen_pb(g, x, y, size - q4, size); gen_pb(g, x + size - q4, y, q4, size); break;
    }
    if (!skip && rbelow(g, 10) < 7) {
      int lm[4] = {0, 0, 0, 0};
      gen_tt(g, x, y, log2, 0, 0, lm, 0, qp, 0);
    } else {
      m355_tu* tu = (m355_tu*)vpush(&g->tus, sizeof(m355_tu));   /* no transform tree: one leaf of CU size (deblock.cc:39) */
      tu->x = (uint16_t)x; tu->y = (uint16_t)y; tu->log2_size = (uint8_t)log2;
    }
  }
}

static void gen_cq(gen* g, int x, int y, int log2, int target)
{
  /* coding quadtree with the implicit split at the picture boundary (slice.cc:4650) */
  const int size = 1 << log2;
  if (x >= g->cfg->width || y >= g->cfg->height) return;
  const int fits = x + size <= g->cfg->width && y + size <= g->cfg->height;
  if (log2 > 3 && (!fits || log2 > target)) {
    const int h = size / 2;
    gen_cq(g, x, y, log2 - 1, target); gen_cq(g, x + h, y, log2 - 1, target);
    gen_cq(g, x, y + h, log2 - 1, target); gen_cq(g, x + h, y + h, log2 - 1, target);
  } else if (fits) gen_cu(g, x, y, log2);
}

typedef struct m355_synth_out {
  m355_picture pic;
  void* owned[16];
} m355_synth_out;

__attribute__((visibility("default"))) int m355_synth_picture(const m355_synth_cfg* cfg, m355_synth_out* out)
{
  gen g; memset(&g, 0, sizeof(g));
  g.cfg = cfg; g.s = cfg->seed ? cfg->seed : 0xC5C5C5C5u;
  const int cs = 1 << cfg->log2_ctb;
  g.ctbW = (cfg->width + cs - 1) / cs; g.ctbH = (cfg->height + cs - 1) / cs;
  if (cfg->width % 8 || cfg->height % 8 || cfg->tile_cols < 1 || cfg->tile_rows < 1 || cfg->tile_cols > M355_MAX_TILE_COLS ||
      cfg->tile_rows > M355_MAX_TILE_ROWS || cfg->tile_cols > g.ctbW || cfg->tile_rows > g.ctbH)
    return M355_ERR_INVALID;
  memset(out, 0, sizeof(*out));
  m355_picture* pic = &out->pic;
  m355_pic_params* pp = &pic->pp;
  g.cf = cfg->chroma_format == 0 ? 1 : (cfg->chroma_format == 4 ? 0 : cfg->chroma_format);
  pp->width = cfg->width; pp->height = cfg->height; pp->chroma_format_idc = (uint8_t)g.cf;
  pp->bit_depth_luma = pp->bit_depth_chroma = (uint8_t)cfg->bit_depth;
  pp->log2_ctb_size = (uint8_t)cfg->log2_ctb; pp->log2_min_tb_size = 2; pp->log2_min_cb_size = 3;
  pp->pic_cb_qp_offset = 1; pp->pic_cr_qp_offset = -1;
  pp->flags = M355_PF_STRONG_INTRA_SMOOTHING | (cfg->sao ? M355_PF_SAO_ENABLED : 0) | (cfg->deblock ? M355_PF_DEBLOCK_ENABLED : 0) |
              (cfg->lf_across_tiles ? M355_PF_LF_ACROSS_TILES : 0) |
              ((cfg->features & M355_SYN_CONSTRAINED_INTRA) ? M355_PF_CONSTRAINED_INTRA_PRED : 0) |
              ((cfg->features & M355_SYN_SCALING_LIST) ? M355_PF_SCALING_LIST : 0) |
              ((cfg->features & M355_SYN_PCM_LOOP_FILTER_DISABLE) ? M355_PF_PCM_LOOP_FILTER_DISABLE : 0) |
              (((cfg->features & M355_SYN_CROSS_COMPONENT) && g.cf == 3) ? M355_PF_CROSS_COMPONENT_PRED : 0) |
              ((cfg->features & M355_SYN_RDPCM) ? M355_PF_IMPLICIT_RDPCM : 0) |
              ((cfg->features & M355_SYN_ROTATE) ? M355_PF_TRANSFORM_SKIP_ROTATION : 0);
  pp->num_tile_cols = (uint8_t)cfg->tile_cols; pp->num_tile_rows = (uint8_t)cfg->tile_rows;
  for (int i = 0; i <= cfg->tile_cols; i++) pp->col_bd[i] = (uint16_t)((i * g.ctbW) / cfg->tile_cols);   /* uniform spacing (pps.cc) */
  for (int i = 0; i <= cfg->tile_rows; i++) pp->row_bd[i] = (uint16_t)((i * g.ctbH) / cfg->tile_rows);
  pic->dst_frame = -1;
  for (int i = 0; i < M355_MAX_REF_FRAMES; i++) pic->ref_frames[i] = -1;

  m355_slice* sl = (m355_slice*)vpush(&g.slices, sizeof(m355_slice));
  sl->slice_addr_rs = 0; sl->beta_offset = (int8_t)(2 * rrange(&g, -2, 2)); sl->tc_offset = (int8_t)(2 * rrange(&g, -2, 2));
  sl->flags = M355_SF_LF_ACROSS_SLICES | M355_SF_SAO_LUMA | M355_SF_SAO_CHROMA;
  /* more slices: start points in tile-scan order (tile starts when the picture has tiles), random per-slice flags */
  const int nctb = g.ctbW * g.ctbH, ntiles = cfg->tile_cols * cfg->tile_rows;
  uint8_t* slice_start = (uint8_t*)calloc((size_t)nctb + 1, 1);
  if (cfg->n_slices > 1) {
    if (ntiles > 1) {
      int ts = 0;
      for (int ty = 0; ty < cfg->tile_rows; ty++)
        for (int tx = 0; tx < cfg->tile_cols; tx++) {
          if (ts && rbelow(&g, ntiles) < cfg->n_slices) slice_start[ts] = 1;
          ts += (pp->row_bd[ty + 1] - pp->row_bd[ty]) * (pp->col_bd[tx + 1] - pp->col_bd[tx]);
        }
    } else {
      for (int k = 1; k < cfg->n_slices; k++) slice_start[1 + rbelow(&g, nctb - 1 > 0 ? nctb - 1 : 1)] = nctb > 1;
    }
  }
  int ts_run = 0, cur_slice = 0;

  g.ctbs.p = calloc((size_t)g.ctbW * g.ctbH, sizeof(m355_ctb)); g.ctbs.n = g.ctbs.cap = (size_t)g.ctbW * g.ctbH; g.ctbs.esz = sizeof(m355_ctb);
  m355_ctb* ctbs = (m355_ctb*)g.ctbs.p;
  for (int ty = 0; ty < cfg->tile_rows; ty++)
    for (int tx = 0; tx < cfg->tile_cols; tx++)
      for (int cy = pp->row_bd[ty]; cy < pp->row_bd[ty + 1]; cy++)
        for (int cx = pp->col_bd[tx]; cx < pp->col_bd[tx + 1]; cx++) {
          m355_ctb* ctb = &ctbs[cy * g.ctbW + cx];
          if (slice_start[ts_run]) {
            m355_slice* ns = (m355_slice*)vpush(&g.slices, sizeof(m355_slice));
            ns->slice_addr_rs = cy * g.ctbW + cx;
            ns->beta_offset = (int8_t)(2 * rrange(&g, -3, 3)); ns->tc_offset = (int8_t)(2 * rrange(&g, -3, 3));
            ns->flags = (uint8_t)((rbelow(&g, 4) == 0 ? M355_SF_DEBLOCK_DISABLED : 0) | (rbelow(&g, 2) ? M355_SF_LF_ACROSS_SLICES : 0) |
                                  (rbelow(&g, 5) ? M355_SF_SAO_LUMA : 0) | (rbelow(&g, 5) ? M355_SF_SAO_CHROMA : 0));
            cur_slice = (int)g.slices.n - 1;
          }
          ts_run++;
          ctb->slice_idx = (uint16_t)cur_slice;
          ctb->ib_start = (uint32_t)g.ibs.n;
          g.ctb_has_bypass = 0;
          const int target = cfg->fixed_cu_log2 ? cfg->fixed_cu_log2 : 3 + rbelow(&g, cfg->log2_ctb - 2);
          gen_cq(&g, cx * cs, cy * cs, cfg->log2_ctb, target > cfg->log2_ctb ? cfg->log2_ctb : target);
          ctb->ib_count = (uint32_t)g.ibs.n - ctb->ib_start;
          if (g.ctb_has_bypass) ctb->flags |= M355_CTBF_HAS_PCM_OR_BYPASS;
          if (cfg->sao) {
            const int tl = rbelow(&g, 6), tc = rbelow(&g, 6); /* off, band, edge x4 */
            const int typ_l = tl == 0 ? 0 : (tl == 1 ? 1 : 2), typ_c = tc == 0 ? 0 : (tc == 1 ? 1 : 2);
            ctb->sao_type = (uint8_t)(typ_l | (typ_c << 2) | (typ_c << 4));
            const int cl = tl >= 2 ? tl - 2 : 0, cc = tc >= 2 ? tc - 2 : 0;
            ctb->sao_eo_class = (uint8_t)(cl | (cc << 2) | (cc << 4));
            const int lim = cfg->bit_depth <= 8 ? 7 : 31;
            for (int k = 0; k < 3; k++) {
              ctb->sao_band_pos[k] = (uint8_t)rbelow(&g, 32);
              for (int j = 0; j < 4; j++) ctb->sao_offset[k][j] = (int8_t)rrange(&g, -lim, lim);
            }
          }
        }

  if (cfg->features & M355_SYN_ROTATE) {
    /* transform_skip_rotation (transform.cc:400-402): every 4x4 skip / bypass block whose CU is intra — where the reference
       looks the prediction mode up at the block's (xT,yT) in COMPONENT samples read as a luma position, which for subsampled
       chroma is another CU; restated literally, so the lists carry what the reference would do */
    const int wcb = (cfg->width + 7) >> 3, hcb = (cfg->height + 7) >> 3;
    uint8_t* is_intra = (uint8_t*)calloc((size_t)wcb * hcb, 1);
    const m355_cu* cus = (const m355_cu*)g.cus.p;
    for (size_t i = 0; i < g.cus.n; i++) {
      if (cus[i].pred_mode != 0) continue;
      const int n8 = 1 << (cus[i].log2_size - 3);
      for (int yy = 0; yy < n8; yy++) memset(is_intra + (size_t)((cus[i].y >> 3) + yy) * wcb + (cus[i].x >> 3), 1, (size_t)n8);
    }
    m355_rb* r4 = (m355_rb*)g.rbs[0].p;
    for (size_t i = 0; i < g.rbs[0].n; i++)
      if ((r4[i].kind == M355_RK_SKIP || r4[i].kind == M355_RK_BYPASS) && is_intra[(size_t)(r4[i].y >> 3) * wcb + (r4[i].x >> 3)]) r4[i].flags |= M355_RBF_ROTATE;
    free(is_intra);
  }

  if (cfg->features & M355_SYN_RDPCM) {
    /* implicit_rdpcm_enabled_flag: angular prediction inside cu_transquant_bypass CUs runs without the boundary filter
       (intrapred.cc:306-308) — the bypass flag is looked up at the block's position in COMPONENT samples, as the reference does */
    const int wcb = (cfg->width + 7) >> 3, hcb = (cfg->height + 7) >> 3;
    uint8_t* is_bypass = (uint8_t*)calloc((size_t)wcb * hcb, 1);
    const m355_cu* cus = (const m355_cu*)g.cus.p;
    for (size_t i = 0; i < g.cus.n; i++) {
      if (!(cus[i].flags & M355_CUF_TRANSQUANT_BYPASS)) continue;
      const int n8 = 1 << (cus[i].log2_size - 3);
      for (int yy = 0; yy < n8; yy++) memset(is_bypass + (size_t)((cus[i].y >> 3) + yy) * wcb + (cus[i].x >> 3), 1, (size_t)n8);
    }
    m355_ib* ibs = (m355_ib*)g.ibs.p;
    for (size_t i = 0; i < g.ibs.n; i++)
      if (!(ibs[i].flags & M355_IBF_PCM) && is_bypass[(size_t)(ibs[i].y >> 3) * wcb + (ibs[i].x >> 3)]) ibs[i].flags |= M355_IBF_DISABLE_BOUNDARY_FILTER;
    free(is_bypass);
  }

  /* concatenate the four size bins */
  size_t nrb = 0;
  for (int s = 0; s < 4; s++) { pic->rb_count[s] = (int32_t)g.rbs[s].n; nrb += g.rbs[s].n; }
  m355_rb* rbs = (m355_rb*)malloc((nrb ? nrb : 1) * sizeof(m355_rb));
  size_t o = 0;
  for (int s = 0; s < 4; s++) { if (g.rbs[s].n) memcpy(rbs + o, g.rbs[s].p, g.rbs[s].n * sizeof(m355_rb)); o += g.rbs[s].n; free(g.rbs[s].p); }

  pic->n_slices = (int32_t)g.slices.n; pic->n_ctbs = (int32_t)g.ctbs.n; pic->n_cus = (int32_t)g.cus.n; pic->n_tus = (int32_t)g.tus.n;
  pic->n_pbs = (int32_t)g.pbs.n; pic->n_wts = (int32_t)g.wts.n; pic->n_ibs = (int32_t)g.ibs.n;
  pic->n_coeffs = (uint32_t)g.coeffs.n; pic->n_pcm = (uint32_t)g.pcm.n; pic->res_len = g.res_len;
  pic->slices = (const m355_slice*)g.slices.p; pic->ctbs = ctbs; pic->cus = (const m355_cu*)g.cus.p; pic->tus = (const m355_tu*)g.tus.p;
  pic->pbs = (const m355_pb*)g.pbs.p; pic->wts = (const m355_wt*)g.wts.p; pic->rbs = rbs; pic->ibs = (const m355_ib*)g.ibs.p;
  pic->coeffs = (const uint32_t*)g.coeffs.p; pic->pcm = (const uint16_t*)g.pcm.p; pic->scaling_factors = NULL;
  uint8_t* sf = NULL;
  if (cfg->features & M355_SYN_SCALING_LIST) {   /* ScalingFactor tables [sizeId][matrixID][y][x] (sps.h:58-65): flat 16 with random relief */
    const int nsf = 6 * (16 + 64 + 256 + 1024);
    sf = (uint8_t*)malloc((size_t)nsf);
    for (int i = 0; i < nsf; i++) sf[i] = (uint8_t)(rbelow(&g, 4) == 0 ? rrange(&g, 1, 255) : rrange(&g, 8, 40));
    pic->scaling_factors = sf;
  }
  free(slice_start);
  void* own[] = {g.slices.p, g.ctbs.p, g.cus.p, g.tus.p, g.pbs.p, g.wts.p, rbs, g.ibs.p, g.coeffs.p, sf, g.pcm.p};
  for (int i = 0; i < 11; i++) out->owned[i] = own[i];
  return M355_OK;
}

__attribute__((visibility("default"))) void m355_synth_free(m355_synth_out* out)
{
  for (int i = 0; i < 16; i++) { free(out->owned[i]); out->owned[i] = NULL; }
}

/* reference plane content: smoothed (3x3 box) noise + gradient, tight w x h samples */
__attribute__((visibility("default"))) void m355_synth_ref_plane(uint32_t seed, int w, int h, int bit_depth, void* dst)
{
  uint32_t s = seed ? seed : 1;
  uint16_t* n = (uint16_t*)malloc((size_t)w * h * 2);
  const int maxv = (1 << bit_depth) - 1;
  for (size_t i = 0; i < (size_t)w * h; i++) { s ^= s << 13; s ^= s >> 17; s ^= s << 5; n[i] = (uint16_t)(s % (uint32_t)(maxv + 1)); }
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      int acc = 0, cnt = 0;
      for (int dy = -1; dy <= 1; dy++)
        for (int dx = -1; dx <= 1; dx++) {
          const int yy = y + dy, xx = x + dx;
          if (yy < 0 || xx < 0 || yy >= h || xx >= w) continue;
          acc += n[(size_t)yy * w + xx]; cnt++;
        }
      int v = (acc / cnt) / 2 + ((x + y) * maxv) / (2 * (w + h));
      if (v > maxv) v = maxv;
      if (bit_depth <= 8) ((uint8_t*)dst)[(size_t)y * w + x] = (uint8_t)v; else ((uint16_t*)dst)[(size_t)y * w + x] = (uint16_t)v;
    }
  free(n);
}

/* ---- stand-in for a recorder that writes its lists IN PLACE (m355_arena_begin): copies a generated picture's lists into the
 * arena pointers of `dst` (all plain memory traffic, spread over `threads` OpenMP threads like parser threads would be) and
 * fills in dst's counts / parameters.  Benchmarks only. ---- */
#include <omp.h>
__attribute__((visibility("default"))) void m355_synth_fill_arena(const m355_picture* src, const m355_arena_caps* caps, m355_picture* dst, int threads)
{
  struct { void* d; const void* s; size_t n; } job[16];
  int nj = 0;
#define JOB(D, S, N) do { job[nj].d = (void*)(D); job[nj].s = (S); job[nj].n = (N); nj++; } while (0)
  JOB(dst->slices, src->slices, sizeof(m355_slice) * (size_t)src->n_slices);
  JOB(dst->ctbs, src->ctbs, sizeof(m355_ctb) * (size_t)src->n_ctbs);
  JOB(dst->cus, src->cus, sizeof(m355_cu) * (size_t)src->n_cus);
  JOB(dst->tus, src->tus, sizeof(m355_tu) * (size_t)src->n_tus);
  JOB(dst->pbs, src->pbs, sizeof(m355_pb) * (size_t)src->n_pbs);
  JOB(dst->wts, src->wts, sizeof(m355_wt) * (size_t)src->n_wts);
  size_t o = 0;
  for (int b = 0; b < 4; b++) { JOB(caps->rb_bin[b], src->rbs + o, sizeof(m355_rb) * (size_t)src->rb_count[b]); o += (size_t)src->rb_count[b]; }
  JOB(dst->ibs, src->ibs, sizeof(m355_ib) * (size_t)src->n_ibs);
  JOB(dst->coeffs, src->coeffs, 4 * (size_t)src->n_coeffs);
  JOB(dst->pcm, src->pcm, 2 * (size_t)src->n_pcm);
  if (src->scaling_factors && dst->scaling_factors) JOB(dst->scaling_factors, src->scaling_factors, 6 * (16 + 64 + 256 + 1024));
#undef JOB
  const size_t CH = (size_t)1 << 18;
  size_t nchunks[16], tot = 0;
  for (int j = 0; j < nj; j++) { nchunks[j] = (job[j].n + CH - 1) / CH; tot += nchunks[j]; }
  if (threads < 1) threads = 1;
#pragma omp parallel for num_threads(threads) schedule(dynamic, 4)
  for (long long g = 0; g < (long long)tot; g++) {
    size_t k = (size_t)g; int j = 0;
    while (k >= nchunks[j]) { k -= nchunks[j]; j++; }
    const size_t b = k * CH, e = b + CH < job[j].n ? b + CH : job[j].n;
    memcpy((char*)job[j].d + b, (const char*)job[j].s + b, e - b);
  }
  dst->pp = src->pp;
  dst->n_slices = src->n_slices; dst->n_ctbs = src->n_ctbs; dst->n_cus = src->n_cus; dst->n_tus = src->n_tus; dst->n_pbs = src->n_pbs;
  dst->n_wts = src->n_wts; dst->n_ibs = src->n_ibs; dst->n_coeffs = src->n_coeffs; dst->n_pcm = src->n_pcm; dst->res_len = src->res_len;
  for (int b = 0; b < 4; b++) dst->rb_count[b] = src->rb_count[b];
  if (!src->scaling_factors) dst->scaling_factors = NULL;
}
/* counts / parameters only (the arena still holds the lists of an earlier fill) */
__attribute__((visibility("default"))) void m355_synth_fill_arena_header(const m355_picture* src, m355_picture* dst)
{
  dst->pp = src->pp;
  dst->n_slices = src->n_slices; dst->n_ctbs = src->n_ctbs; dst->n_cus = src->n_cus; dst->n_tus = src->n_tus; dst->n_pbs = src->n_pbs;
  dst->n_wts = src->n_wts; dst->n_ibs = src->n_ibs; dst->n_coeffs = src->n_coeffs; dst->n_pcm = src->n_pcm; dst->res_len = src->res_len;
  for (int b = 0; b < 4; b++) dst->rb_count[b] = src->rb_count[b];
  if (!src->scaling_factors) dst->scaling_factors = NULL;
}
