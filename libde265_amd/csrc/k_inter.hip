/*
 * k_inter.hip — motion-compensated inter prediction for all prediction blocks of a picture.
 *
 * Replaces, per PB, the whole of generate_inter_prediction_samples (motion.cc:288-730):
 *   mc_luma / mc_chroma incl. the picture-edge coordinate clamp (motion.cc:48-282),
 *   put_hevc_qpel / put_hevc_epel* (fallback-motion.cc:492-636, 305-415, 431-485, 262-302),
 *   put_unweighted_pred / put_weighted_pred_avg / put_weighted_pred / put_weighted_bipred
 *   (fallback-motion.cc:33-256).
 *
 * Main kernel (4:2:0 and monochrome), k_inter_jobs — register-resident, no LDS data staging:
 *   k_meta_pb expands every PB into JOBS of 4 luma columns x 8 rows (+ the matching 2x4 Cb and Cr
 *   samples); one LANE owns one job, consecutive lanes own horizontally adjacent jobs of the same
 *   PB, so a wave's loads and stores are row-contiguous wherever a PB is >= 8 wide.  Per window row
 *   a lane loads the 12 samples it needs with one unaligned 16-byte + one 8-byte load (overlap with
 *   the neighbour lane is served by the L1), keeps them as packed 16-bit pairs and evaluates the
 *   8-tap (4-tap) horizontal filter with v_dot2c_i32_i16 — two taps per issue, "odd" output columns
 *   use a tap set shifted by one sample — truncating to int16 exactly where the reference stores
 *   its mcbuffer.  The 16-bit intermediates of the 15 (7) window rows stay in registers as
 *   (row 2k, row 2k+1) pairs, the vertical filter is the same dot2 scheme, and the weighted
 *   combination writes 8-byte row segments straight to the picture.  Every lane carries its own
 *   MV phases / taps / weights (VGPRs): PBs of any size and mix share a wave with no divergence
 *   except the rare picture-edge path (per-sample clamped loads, motion.cc:141-159).
 *   16-bit planes with bit depth 16 use the identity sum(t*s) = sum(t*(s-32768)) + 32768*64 so
 *   samples fit the signed dot2 operands.
 * Generic kernel (4:2:2 / 4:4:4), k_inter_generic: one wavefront per PB, 16x16 tiles through LDS.
 *
 * Roofline: HBM/L2-bound; algorithmic bytes per PB and list = [(w+7)(h+7)+2(w/2+3)(h/2+3)]*B read,
 * w*h*1.5*B written.
 */
#include <stdlib.h>
#include "k_common.h"

__constant__ int8_t c_qpel_taps[4][8] = {{0, 0, 0, 64, 0, 0, 0, 0},
                                         {-1, 4, -10, 58, 17, -5, 1, 0},
                                         {-1, 4, -11, 40, 40, -11, 4, -1},
                                         {0, 1, -5, 17, 58, -10, 4, -1}};
__constant__ int8_t c_epel_taps[8][4] = {{0, 64, 0, 0},   {-2, 58, 10, -2}, {-4, 54, 16, -2}, {-6, 46, 28, -4},
                                         {-4, 36, 36, -4}, {-4, 28, 46, -6}, {-2, 16, 54, -4}, {-2, 10, 58, -2}};

#define WIN_PITCH 24 /* >= 16+7 */
#define WIN_ROWS 23

template <class PIX>
__global__ void __launch_bounds__(256) k_inter_generic(DevPic p)
{
  M355_GATE(p);
  __shared__ uint16_t s_win[4][WIN_ROWS * WIN_PITCH];
  __shared__ int16_t s_tmp[4][WIN_ROWS * 16];

  /* everything derived from the PB record is wave-uniform: tell the compiler (readfirstlane) so the
     record, the MV phases and the filter taps live in SGPRs and the loops use scalar branches */
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  const int pbi = blockIdx.x * 4 + wave;
  if (pbi >= p.n_pbs) return; /* wave-uniform */
  const m355_pb pb = p.pbs[pbi];
  uint16_t* win = s_win[wave];
  int16_t* tmp = s_tmp[wave];
  const int nc = p.pp.chroma_format_idc ? 3 : 1;
  const bool mc0 = pb.flags & M355_PBF_MC_L0, mc1 = pb.flags & M355_PBF_MC_L1;
  const bool weighted = pb.flags & M355_PBF_WEIGHTED;

  for (int c = 0; c < nc; c++) {
    const int bd = c ? p.pp.bit_depth_chroma : p.pp.bit_depth_luma;
    const int W = c ? pb.w / p.sw : pb.w, H = c ? pb.h / p.sh : pb.h;
    const int xP = c ? pb.x / p.sw : pb.x, yP = c ? pb.y / p.sh : pb.y;
    const int ntaps = c ? 4 : 8, before = c ? 1 : 3;
    const int shift1 = bd - 8, shift3 = max(2, 14 - bd);
    PIX* dstp = (PIX*)p.plane[c];
    const int dstride = p.stride[c];
    const int pw = p.pw[c], ph = p.ph[c];

    for (int ty = 0; ty < H; ty += 16)
      for (int tx = 0; tx < W; tx += 16) {
        const int tw = min(16, W - tx), th = min(16, H - ty);
        int pred[2][4];
#pragma unroll
        for (int l = 0; l < 2; l++) {
          const bool use = l ? mc1 : mc0;
          if (!use) continue;
          if (pb.flags & (M355_PBF_FILL_L0 << l)) { /* motion.cc:362-376 */
#pragma unroll
            for (int k = 0; k < 4; k++) pred[l][k] = 1 << 13;
            continue;
          }
          const DevRef* ref = &p.refs[pb.ref_slot[l]];
          const PIX* rp = (const PIX*)ref->plane[c];
          const int rstride = ref->stride[c];
          int mvx = pb.mv[l][0], mvy = pb.mv[l][1];
          int xf, yf, xi, yi;
          if (c == 0) { xf = mvx & 3; yf = mvy & 3; xi = xP + (mvx >> 2); yi = yP + (mvy >> 2); }
          else { mvx *= 2 / p.sw; mvy *= 2 / p.sh; xf = mvx & 7; yf = mvy & 7; xi = xP + (mvx >> 3); yi = yP + (mvy >> 3); }
          int tpx[8], tpy[8];
#pragma unroll
          for (int t = 0; t < 8; t++) {
            tpx[t] = c == 0 ? c_qpel_taps[xf][t] : (t < 4 ? c_epel_taps[xf][t] : 0);
            tpy[t] = c == 0 ? c_qpel_taps[yf][t] : (t < 4 ? c_epel_taps[yf][t] : 0);
          }
          /* 1. reference window -> LDS (coordinate clamp = picture-edge padding, motion.cc:141-159);
                32 lanes per window row, two rows per step: no integer division anywhere */
          const int ww = tw + ntaps - 1, wh = th + ntaps - 1;
          {
            const int wx = lane & 31, wy0 = lane >> 5;
            const int xa = d_clip3(0, pw - 1, xi + tx + wx - before);
            if (wx < ww)
              for (int wy = wy0; wy < wh; wy += 2) {
                const int ya = d_clip3(0, ph - 1, yi + ty + wy - before);
                win[wy * WIN_PITCH + wx] = rp[ya * rstride + xa];
              }
          }
          wave_sync();
          /* 2. horizontal taps -> int16 (fallback-motion.cc:512-565 / 350-375); 16 lanes per row */
          {
            const int x = lane & 15, r0 = lane >> 4;
            if (x < tw)
              for (int r = r0; r < wh; r += 4) {
                int v;
                if (xf == 0) v = win[r * WIN_PITCH + x + before];
                else {
                  int s = 0;
                  if (c == 0) {
#pragma unroll
                    for (int k = 0; k < 8; k++) s += tpx[k] * win[r * WIN_PITCH + x + k];
                  } else {
#pragma unroll
                    for (int k = 0; k < 4; k++) s += tpx[k] * win[r * WIN_PITCH + x + k];
                  }
                  v = s >> shift1;
                }
                tmp[r * 16 + x] = (int16_t)v;
              }
          }
          wave_sync();
          /* 3. vertical taps -> registers (fallback-motion.cc:573-626 / 381-404); output k = rows 4k.. */
          const int vshift = (xf == 0) ? shift1 : 6;
#pragma unroll
          for (int k = 0; k < 4; k++) {
            const int x = lane & 15, y = 4 * k + (lane >> 4);
            int v = 0;
            if (x < tw && y < th) {
              if (yf == 0) {
                v = tmp[(y + before) * 16 + x];
                if (xf == 0) v = (int)((unsigned)v & 0xFFFF) << shift3; /* full-pel: ref << shift3 */
              } else {
                int s = 0;
                if (c == 0) {
#pragma unroll
                  for (int t = 0; t < 8; t++) s += tpy[t] * tmp[(y + t) * 16 + x];
                } else {
#pragma unroll
                  for (int t = 0; t < 4; t++) s += tpy[t] * tmp[(y + t) * 16 + x];
                }
                v = s >> vshift;
              }
            }
            pred[l][k] = (int16_t)v;
          }
          wave_sync(); /* win/tmp are reused by the next list / tile */
        }
        /* 4. weighted write-back (fallback-motion.cc:33-256, selection motion.cc:493-688) */
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const int x = lane & 15, y = 4 * k + (lane >> 4);
          if (x >= tw || y >= th) continue;
          int v;
          if (mc0 && mc1) {
            if (weighted) {
              const m355_wt w0 = p.wts[pb.wt_idx[0]], w1 = p.wts[pb.wt_idx[1]];
              const int log2WD = c ? w0.log2wd_chroma : w0.log2wd_luma;
              const int rnd = (int)((unsigned)(w0.o[c] + w1.o[c] + 1) << log2WD);
              v = (pred[0][k] * w0.w[c] + pred[1][k] * w1.w[c] + rnd) >> (log2WD + 1);
            } else {
              const int shift2 = max(3, 15 - bd);
              v = (pred[0][k] + pred[1][k] + (1 << (shift2 - 1))) >> shift2;
            }
          } else {
            const int l = mc0 ? 0 : 1;
            if (weighted) {
              const m355_wt w = p.wts[pb.wt_idx[l]];
              const int log2WD = c ? w.log2wd_chroma : w.log2wd_luma;
              v = ((pred[l][k] * w.w[c] + (1 << (log2WD - 1))) >> log2WD) + w.o[c];
            } else {
              v = (pred[l][k] + (1 << (shift3 - 1))) >> shift3;
            }
          }
          dstp[(yP + ty + y) * dstride + xP + tx + x] = (PIX)d_clip_bd(v, bd);
        }
      }
  }
}


/* ================================================================================================
 * job kernel
 * ============================================================================================== */

#ifndef M355_INTER_WAVES
#define M355_INTER_WAVES 3   /* waves per SIMD the register budget is sized for (tools/variants.sh sweeps it) */
#endif
#ifndef M355_INTER_BLOCK
#define M355_INTER_BLOCK 256   /* lanes (jobs) per workgroup */
#endif
#define QT_STRIDE 9
#define ET_STRIDE 5
__device__ __forceinline__ unsigned d_pack16(int lo, int hi) { return ((unsigned)lo & 0xFFFFu) | ((unsigned)hi << 16); }
/* the write-back weights of one component as ONE formula: all four reference forms (fallback-motion.cc:33-256, selection motion.cc:493-688) are
 * instances of ((a*w0 + b*w1 + rnd) >> sh) + o with the same int32 arithmetic:
 *   unweighted uni : w0 1, w1 0, rnd 1<<(shift3-1),        sh shift3,   o 0      (put_unweighted_pred)
 *   average        : w0 1, w1 1, rnd 1<<(shift2-1),        sh shift2,   o 0      (put_weighted_pred_avg)
 *   weighted uni   : w0,   w1 0, rnd 1<<(log2WD-1),        sh log2WD,   o o0     (put_weighted_pred)
 *   weighted bi    : w0,   w1,   rnd (o0+o1+1)<<log2WD,    sh log2WD+1, o 0      (put_weighted_bipred)
 * so the per-lane mode costs no branch. */
struct WtSel { int w0, w1, rnd, sh, o; };

/* ================================================================================================
 * lean filters (bit depths <= 12, i.e. every 8-bit plane and the 9..12-bit uint16 planes): the job classes whose windows lie
 * INSIDE the reference picture (k_meta_pb sorts the others into the EDGE range).
 *
 *  * The reference's special cases fold into the general filter exactly (fallback-motion.cc:431-485 full-pel, :512-565 / :573-626 the
 *    xFrac == 0 / yFrac == 0 copies): phase 0 is the tap row {0,0,0,64,0,0,0,0} ({0,64,0,0} chroma), so the H pass gives
 *    (64 s) >> (bd - 8) = s << (14 - bd) (<= 16 380: an int16, nothing lost for bd <= 14), the V pass with shift 6 gives
 *    (sum t (s << (14 - bd))) >> 6 = (sum t s) >> (bd - 8) for xFrac == 0, (64 h) >> 6 = h for yFrac == 0 and s << (14 - bd) =
 *    s << shift3 (bd <= 12) for both: no selects, no second data path.
 *  * The right shift of a pass is made 8 by scaling its taps (H: x 2^(16 - bd), <= 88 * 128; V: x 4; all products and sums stay far
 *    inside int32), so that shift + truncation to int16 (the reference's int16 mcbuffer / predSamples stores) + packing two results
 *    into a register pair is ONE v_perm_b32 (d_pack_mid16) instead of two shifts and a pack.
 *  * Window rows are loaded from the dword-aligned address at or below their first sample (k_asm.h d_ldg*: a misaligned vector load
 *    costs the texture addresser 4x) and NOT shifted into place: the lane picks the tap registers that match its window's phase
 *    d = x & 1 instead (16-bit planes: T0 / T1 below; 20 dot2 per row instead of 18 + 6 funnel shifts).
 *  * 8-bit planes: v_dot4_i32_i8 on the bytes as loaded (3 funnel shifts per 16-byte row, XOR 0x80 makes them signed operands,
 *    + 128 * 64 in the accumulator): 11 dot4 per luma row instead of 12 unpacks + 18 dot2.
 *  * No coordinate clamps: address = first row + r * pitch.
 * ============================================================================================== */
#ifndef M355_INTER_PIPE
#define M355_INTER_PIPE 2   /* luma window rows: 0 = all 15 requested at once (hipcc's own order), D = D row pairs ahead of the arithmetic */
#endif
#ifndef M355_INTER_EDGE_DEPTH
#define M355_INTER_EDGE_DEPTH 4   /* EDGE jobs of 16-bit planes: row pairs requested ahead */
#endif
#define QL_STRIDE 12   /* 16-bit planes: [xf][d][12] = T0[5] T1[5] (2 spare);  8-bit planes: [xf][12] = W[j][3], j = output column */
#define CL_STRIDE 8    /* 16-bit planes: [xf][d][8]  = U0[3] U1[3] (2 spare);  8-bit planes: [xf][4]  = C0 C1 C2 (1 spare) */
/* The tap tables of the lean filters are made ONCE per plane type and bit-depth pair, on the host (runtime.hip keeps them in device
 * memory, DevPic.inter_tabs), and a workgroup copies them into LDS with one coalesced load beside its first record loads — built per
 * workgroup from immediates they cost a wave ~300 vector instructions, a sixth of a job's arithmetic.
 *   [QL_MAIN 96] H luma of the classes inside the picture   : 16-bit planes [xf][d][12] = T0[5] T1[5];  8-bit planes [xf][12] = W[j][3]
 *   [QV 36]      V luma x 4                                  : [yf][9] = E0..E3 O0..O4
 *   [CL_MAIN 128] H chroma                                   : 16-bit planes [xf][d][8] = U0[3] U1[3];   8-bit planes [xf][4] = C0 C1 C2
 *   [CV 40]      V chroma x 4                                : [yf][5] = E0 E1 O0 O1 O2
 */
#define LT_QL 0
#define LT_QV 96
#define LT_CL 132
#define LT_CV 260
#define LT_WORDS 300
static int h_qtap(int f, int i) { static const int8_t t[4][8] = {{0, 0, 0, 64, 0, 0, 0, 0}, {-1, 4, -10, 58, 17, -5, 1, 0}, {-1, 4, -11, 40, 40, -11, 4, -1}, {0, 1, -5, 17, 58, -10, 4, -1}}; return i >= 0 && i < 8 ? t[f][i] : 0; }
static int h_etap(int f, int i) { static const int8_t t[8][4] = {{0, 64, 0, 0}, {-2, 58, 10, -2}, {-4, 54, 16, -2}, {-6, 46, 28, -4}, {-4, 36, 36, -4}, {-4, 28, 46, -6}, {-2, 16, 54, -4}, {-2, 10, 58, -2}}; return i >= 0 && i < 4 ? t[f][i] : 0; }
/* packed pair (t(first), t(first + 1)) x scale of an n-tap filter: E_k = first 2k, O_k = first 2k - 1 */
static uint32_t h_pair(int (*t)(int, int), int f, int first, int scale) { return ((uint32_t)(t(f, first) * scale) & 0xFFFFu) | ((uint32_t)(t(f, first + 1) * scale) << 16); }
void m355_inter_tables(bool bytes, int bd_luma, int bd_chroma, uint32_t* out)
{
  for (int i = 0; i < LT_WORDS; i++) out[i] = 0;
  for (int f = 0; f < 4; f++)
    for (int k = 0; k < QT_STRIDE; k++) out[LT_QV + f * QT_STRIDE + k] = h_pair(h_qtap, f, k < 4 ? 2 * k : 2 * (k - 4) - 1, 4);
  for (int f = 0; f < 8; f++)
    for (int k = 0; k < ET_STRIDE; k++) out[LT_CV + f * ET_STRIDE + k] = h_pair(h_etap, f, k < 2 ? 2 * k : 2 * (k - 2) - 1, 4);
  /* pair form: d = 0: T0 = (E0 E1 E2 E3 0), T1 = (O0 .. O4);  d = 1: T0 = (O0 .. O4), T1 = (0 E0 E1 E2 E3) — pair k of set s holds the taps of the
     window's samples 2k, 2k + 1 for output column d + s (= the sample index of that column's first tap) */
  auto pairs = [&](uint32_t* ql, uint32_t* cl, int hs, int hc) {
    for (int f = 0; f < 4; f++)
      for (int d = 0; d < 2; d++)
        for (int set = 0; set < 2; set++)
          for (int k = 0; k < 5; k++) ql[(f * 2 + d) * QL_STRIDE + set * 5 + k] = h_pair(h_qtap, f, 2 * k - (d + set), hs);
    for (int f = 0; f < 8; f++)
      for (int d = 0; d < 2; d++)
        for (int set = 0; set < 2; set++)
          for (int k = 0; k < 3; k++) cl[(f * 2 + d) * CL_STRIDE + set * 3 + k] = h_pair(h_etap, f, 2 * k - (d + set), hc);
  };
  if (!bytes) pairs(out + LT_QL, out + LT_CL, 1 << (16 - bd_luma), 1 << (16 - bd_chroma));
  else {
    for (int f = 0; f < 4; f++)          /* W[j][w]: the 8 taps as bytes at byte offset j of 12 */
      for (int j = 0; j < 4; j++)
        for (int w = 0; w < 3; w++) {
          uint32_t v = 0;
          for (int b = 0; b < 4; b++) v |= ((uint32_t)h_qtap(f, 4 * w + b - j) & 0xFFu) << (8 * b);
          out[LT_QL + f * QL_STRIDE + 3 * j + w] = v;
        }
    for (int f = 0; f < 8; f++) {
      uint32_t c = 0;
      for (int b = 0; b < 4; b++) c |= ((uint32_t)h_etap(f, b) & 0xFFu) << (8 * b);
      out[LT_CL + f * 4] = c; out[LT_CL + f * 4 + 1] = c << 8; out[LT_CL + f * 4 + 2] = c >> 24;
    }
  }
}

/* luma 4x8 block of one list -> packed 14-bit predictions (fallback-motion.cc:492-636 with the folds above) */
/* EDGE: a window that leaves the picture (motion.cc:84-91, 141-159: every sample coordinate clamped): the rows are fetched from clamped
   row indices, the nearest in-range 12 / 16 samples of each with the vector loads of d_load12 and shifted into place with saturation
   (d_shift_sat) as 16-bit pairs — ALL 15 rows requested before the first is used: these jobs are few, what they cost is their LATENCY
   (they used to fetch and filter row pair by row pair: 44 dependent memory round trips per bi-predicted job, the tail of the whole
   launch at 4K: profiles/r05_c_inter_attribution.txt) */
template <class PIX, bool EDGE>
__device__ __forceinline__ void d_mc_luma_lean(const M355_GLOBAL PIX* rp, int rstride, int pw, int ph, int xi, int yi, int xf, int yf,
                                               const unsigned* s_ql, const unsigned* s_qv, unsigned* ext, unsigned pred[8][2])
{
  const int xa = xi - 3;
  unsigned Q[8][4];
  constexpr int DEPTH = M355_INTER_PIPE ? M355_INTER_PIPE : 1, NB = DEPTH + 1;   /* row pairs requested ahead of the one being filtered */
  if (EDGE) {
    if (sizeof(PIX) == 2) {
      const int xb = d_clip3(0, pw - 12, xa), s_ = d_clip3(0, 24, 12 + xa - xb);       /* sample offset of the span in the extended row */
      const unsigned* tl = s_ql + (xf * 2 + (s_ & 1)) * QL_STRIDE;
      unsigned T0[5], T1[5];
#pragma unroll
      for (int k = 0; k < 5; k++) { T0[k] = tl[k]; T1[k] = tl[5 + k]; }
      /* four row pairs requested ahead of the one being filtered (all 15 rows at once would spill) */
      constexpr int DE = M355_INTER_EDGE_DEPTH, NE = DE + 1;
      unsigned G[NE][2][6];
      auto fetch = [&](int k) {
#pragma unroll
        for (int r = 0; r < 2; r++) {
          if (2 * k + r >= 15) break;
          const M355_GLOBAL PIX* q = rp + (ptrdiff_t)d_clip3(0, ph - 1, yi - 3 + 2 * k + r) * rstride + xb;
          d_ldg16(q, G[k % NE][r]); d_ldg8(q + 8, G[k % NE][r] + 4);
        }
      };
#pragma unroll
      for (int k = 0; k < DE; k++) fetch(k);
#pragma unroll
      for (int k = 0; k < 8; k++) {
        if (k + DE < 8) fetch(k + DE);
        int h[2][4];
#pragma unroll
        for (int r = 0; r < 2; r++) {
          if (k == 7 && r == 1) { h[r][0] = h[r][1] = h[r][2] = h[r][3] = 0; break; }
          const unsigned* g = G[k % NE][r];
          const unsigned L = (g[0] & 0xFFFFu) * 0x10001u, R = (g[5] >> 16) * 0x10001u;
          uint4* e4 = (uint4*)ext;
          e4[0] = make_uint4(L, L, L, L); e4[1] = make_uint4(L, L, g[0], g[1]); e4[2] = make_uint4(g[2], g[3], g[4], g[5]); e4[3] = make_uint4(R, R, R, R);
          ext[16] = R; ext[17] = R;
          unsigned E[6];
#pragma unroll
          for (int i = 0; i < 6; i++) E[i] = ext[(s_ >> 1) + i];
          h[r][0] = d_dot2(E[4], T0[4], d_dot2(E[3], T0[3], d_dot2(E[2], T0[2], d_dot2(E[1], T0[1], d_dot2z(E[0], T0[0])))));
          h[r][1] = d_dot2(E[4], T1[4], d_dot2(E[3], T1[3], d_dot2(E[2], T1[2], d_dot2(E[1], T1[1], d_dot2z(E[0], T1[0])))));
          h[r][2] = d_dot2(E[5], T0[4], d_dot2(E[4], T0[3], d_dot2(E[3], T0[2], d_dot2(E[2], T0[1], d_dot2z(E[1], T0[0])))));
          h[r][3] = d_dot2(E[5], T1[4], d_dot2(E[4], T1[3], d_dot2(E[3], T1[2], d_dot2(E[2], T1[1], d_dot2z(E[1], T1[0])))));
        }
#pragma unroll
        for (int j = 0; j < 4; j++) Q[k][j] = d_pack_mid16((unsigned)h[0][j], (unsigned)h[1][j]);
        M355_PIN_V4_MEM(Q[k][0], Q[k][1], Q[k][2], Q[k][3]);
      }
    } else {
      const int xb = d_clip3(0, pw - 16, xa), s_ = d_clip3(0, 36, 16 + xa - xb);       /* byte offset of the span in the extended row */
      const unsigned* tl = s_ql + xf * QL_STRIDE;
      unsigned W[4][3];
#pragma unroll
      for (int j = 0; j < 4; j++) { W[j][0] = tl[3 * j]; W[j][1] = tl[3 * j + 1]; W[j][2] = tl[3 * j + 2]; }
      const unsigned sh = (unsigned)s_ & 3u;
      unsigned G[15][4];
#pragma unroll
      for (int r = 0; r < 15; r++) d_ldg16(rp + (ptrdiff_t)d_clip3(0, ph - 1, yi - 3 + r) * rstride + xb, G[r]);
#pragma unroll
      for (int k = 0; k < 8; k++) {
        int h[2][4];
#pragma unroll
        for (int r = 0; r < 2; r++) {
          if (k == 7 && r == 1) { h[r][0] = h[r][1] = h[r][2] = h[r][3] = 0; break; }
          const unsigned* g = G[2 * k + r];
          const unsigned L = (g[0] & 0xFFu) * 0x01010101u, R = (g[3] >> 24) * 0x01010101u;
          uint4* e4 = (uint4*)ext;
          e4[0] = make_uint4(L, L, L, L); e4[1] = make_uint4(g[0], g[1], g[2], g[3]); e4[2] = make_uint4(R, R, R, R); e4[3] = make_uint4(R, R, R, R);
          unsigned E[4], A[3];
#pragma unroll
          for (int i = 0; i < 4; i++) E[i] = ext[(s_ >> 2) + i];
#pragma unroll
          for (int i = 0; i < 3; i++) A[i] = __builtin_amdgcn_alignbyte(E[i + 1], E[i], sh) ^ 0x80808080u;
          h[r][0] = d_dot4(A[1], W[0][1], d_dot4k(A[0], W[0][0], 8192));
#pragma unroll
          for (int j = 1; j < 4; j++) h[r][j] = d_dot4(A[2], W[j][2], d_dot4(A[1], W[j][1], d_dot4k(A[0], W[j][0], 8192)));
        }
#pragma unroll
        for (int j = 0; j < 4; j++) Q[k][j] = d_pack_lo16((unsigned)h[0][j], (unsigned)h[1][j]);
      }
    }
  } else if (sizeof(PIX) == 2) {
    const unsigned* tl = s_ql + (xf * 2 + (xa & 1)) * QL_STRIDE;
    unsigned T0[5], T1[5];
#pragma unroll
    for (int k = 0; k < 5; k++) { T0[k] = tl[k]; T1[k] = tl[5 + k]; }
    const M355_GLOBAL PIX* q = rp + (ptrdiff_t)(yi - 3) * rstride + (xa & ~1);
    unsigned S[NB][2][6];
#pragma unroll
    for (int k = 0; k < DEPTH; k++) {
      d_ldg16(q, S[k][0]); d_ldg8(q + 8, S[k][0] + 4); q += rstride;
      d_ldg16(q, S[k][1]); d_ldg8(q + 8, S[k][1] + 4); q += rstride;
    }
#pragma unroll
    for (int k = 0; k < 8; k++) {
      /* the rows of pair k + DEPTH are requested before pair k is filtered; the pin behind a pair's arithmetic (a register
         constraint on its results + a compiler memory barrier) is what holds the order: without it hipcc hoists all 15 rows of
         loads to the top and filters behind one wait for everything — 90 registers of rows in flight and no overlap of a wave's
         own loads with its own arithmetic */
      if (k + DEPTH < 8) {
        d_ldg16(q, S[(k + DEPTH) % NB][0]); d_ldg8(q + 8, S[(k + DEPTH) % NB][0] + 4); q += rstride;
        if (k + DEPTH < 7) { d_ldg16(q, S[(k + DEPTH) % NB][1]); d_ldg8(q + 8, S[(k + DEPTH) % NB][1] + 4); q += rstride; }
      }
      int h[2][4];
#pragma unroll
      for (int r = 0; r < 2; r++) {
        if (k == 7 && r == 1) { h[r][0] = h[r][1] = h[r][2] = h[r][3] = 0; break; }   /* row 15 is never read with a non-zero tap */
        const unsigned* E = S[k % NB][r];
        h[r][0] = d_dot2(E[4], T0[4], d_dot2(E[3], T0[3], d_dot2(E[2], T0[2], d_dot2(E[1], T0[1], d_dot2z(E[0], T0[0])))));
        h[r][1] = d_dot2(E[4], T1[4], d_dot2(E[3], T1[3], d_dot2(E[2], T1[2], d_dot2(E[1], T1[1], d_dot2z(E[0], T1[0])))));
        h[r][2] = d_dot2(E[5], T0[4], d_dot2(E[4], T0[3], d_dot2(E[3], T0[2], d_dot2(E[2], T0[1], d_dot2z(E[1], T0[0])))));
        h[r][3] = d_dot2(E[5], T1[4], d_dot2(E[4], T1[3], d_dot2(E[3], T1[2], d_dot2(E[2], T1[1], d_dot2z(E[1], T1[0])))));
      }
#pragma unroll
      for (int j = 0; j < 4; j++) Q[k][j] = d_pack_mid16((unsigned)h[0][j], (unsigned)h[1][j]);
      if (M355_INTER_PIPE) M355_PIN_V4_MEM(Q[k][0], Q[k][1], Q[k][2], Q[k][3]);
    }
  } else {
    const unsigned* tl = s_ql + xf * QL_STRIDE;
    unsigned W[4][3];
#pragma unroll
    for (int j = 0; j < 4; j++) { W[j][0] = tl[3 * j]; W[j][1] = tl[3 * j + 1]; W[j][2] = tl[3 * j + 2]; }
    const unsigned sh = (unsigned)xa & 3u;
    const M355_GLOBAL PIX* q = rp + (ptrdiff_t)(yi - 3) * rstride + (xa & ~3);
    unsigned S[NB][2][4];
#pragma unroll
    for (int k = 0; k < DEPTH; k++) {
      d_ldg16(q, S[k][0]); q += rstride;
      d_ldg16(q, S[k][1]); q += rstride;
    }
#pragma unroll
    for (int k = 0; k < 8; k++) {
      if (k + DEPTH < 8) {
        d_ldg16(q, S[(k + DEPTH) % NB][0]); q += rstride;
        if (k + DEPTH < 7) { d_ldg16(q, S[(k + DEPTH) % NB][1]); q += rstride; }
      }
      int h[2][4];
#pragma unroll
      for (int r = 0; r < 2; r++) {
        if (k == 7 && r == 1) { h[r][0] = h[r][1] = h[r][2] = h[r][3] = 0; break; }
        const unsigned* E = S[k % NB][r];
        unsigned A[3];
#pragma unroll
        for (int i = 0; i < 3; i++) A[i] = __builtin_amdgcn_alignbyte(E[i + 1], E[i], sh) ^ 0x80808080u;
        h[r][0] = d_dot4(A[1], W[0][1], d_dot4k(A[0], W[0][0], 8192));
#pragma unroll
        for (int j = 1; j < 4; j++) h[r][j] = d_dot4(A[2], W[j][2], d_dot4(A[1], W[j][1], d_dot4k(A[0], W[j][0], 8192)));
      }
#pragma unroll
      for (int j = 0; j < 4; j++) Q[k][j] = d_pack_lo16((unsigned)h[0][j], (unsigned)h[1][j]);
      if (M355_INTER_PIPE) M355_PIN_V4_MEM(Q[k][0], Q[k][1], Q[k][2], Q[k][3]);
    }
  }
  const unsigned* ty = s_qv + yf * QT_STRIDE;
  unsigned YE[4], YO[5];
#pragma unroll
  for (int k = 0; k < 4; k++) YE[k] = ty[k];
#pragma unroll
  for (int k = 0; k < 5; k++) YO[k] = ty[4 + k];
#pragma unroll
  for (int m = 0; m < 4; m++) {
    int ve[4], vo[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      ve[j] = d_dot2(Q[m + 3][j], YE[3], d_dot2(Q[m + 2][j], YE[2], d_dot2(Q[m + 1][j], YE[1], d_dot2z(Q[m][j], YE[0]))));
      vo[j] = d_dot2(Q[m + 4][j], YO[4], d_dot2(Q[m + 3][j], YO[3], d_dot2(Q[m + 2][j], YO[2], d_dot2(Q[m + 1][j], YO[1], d_dot2z(Q[m][j], YO[0])))));
    }
#pragma unroll
    for (int jp = 0; jp < 2; jp++) {
      pred[2 * m][jp] = d_pack_mid16((unsigned)ve[2 * jp], (unsigned)ve[2 * jp + 1]);
      pred[2 * m + 1][jp] = d_pack_mid16((unsigned)vo[2 * jp], (unsigned)vo[2 * jp + 1]);
    }
  }
}

/* chroma 2x4 block of one list and plane (fallback-motion.cc:305-415 / 262-302 with the folds above) */
template <class PIX, bool EDGE>
__device__ __forceinline__ void d_mc_chroma_lean(const M355_GLOBAL PIX* rp, int rstride, int pw, int ph, int xi, int yi, int xf, int yf,
                                                 const unsigned* s_cl, const unsigned* s_cv, unsigned* ext, unsigned pred[4])
{
  const int xa = xi - 1;
  int h[8][2];
  if (EDGE) {
    if (sizeof(PIX) == 2) {
      const int xb = d_clip3(0, pw - 6, xa), s_ = d_clip3(0, 12, 6 + xa - xb);
      const unsigned* tl = s_cl + (xf * 2 + (s_ & 1)) * CL_STRIDE;
      unsigned U0[3], U1[3];
#pragma unroll
      for (int k = 0; k < 3; k++) { U0[k] = tl[k]; U1[k] = tl[3 + k]; }
      unsigned G[7][3];
#pragma unroll
      for (int r = 0; r < 7; r++) d_ldg12(rp + (ptrdiff_t)d_clip3(0, ph - 1, yi - 1 + r) * rstride + xb, G[r]);
#pragma unroll
      for (int r = 0; r < 7; r++) {
        const unsigned* g = G[r];
        const unsigned L = (g[0] & 0xFFFFu) * 0x10001u, R = (g[2] >> 16) * 0x10001u;
        uint4* e4 = (uint4*)ext;
        e4[0] = make_uint4(L, L, L, g[0]); e4[1] = make_uint4(g[1], g[2], R, R); ext[8] = R;
        const unsigned E0 = ext[(s_ >> 1)], E1 = ext[(s_ >> 1) + 1], E2 = ext[(s_ >> 1) + 2];
        h[r][0] = d_dot2(E2, U0[2], d_dot2(E1, U0[1], d_dot2z(E0, U0[0])));
        h[r][1] = d_dot2(E2, U1[2], d_dot2(E1, U1[1], d_dot2z(E0, U1[0])));
      }
    } else {
      const int xb = d_clip3(0, pw - 8, xa), s_ = d_clip3(0, 16, 8 + xa - xb);
      const unsigned* tl = s_cl + xf * 4;
      const unsigned C0 = tl[0], C1 = tl[1], C2 = tl[2];
      const unsigned sh = (unsigned)s_ & 3u;
      unsigned G[7][2];
#pragma unroll
      for (int r = 0; r < 7; r++) d_ldg8(rp + (ptrdiff_t)d_clip3(0, ph - 1, yi - 1 + r) * rstride + xb, G[r]);
#pragma unroll
      for (int r = 0; r < 7; r++) {
        const unsigned* g = G[r];
        const unsigned L = (g[0] & 0xFFu) * 0x01010101u, R = (g[1] >> 24) * 0x01010101u;
        uint4* e4 = (uint4*)ext;
        e4[0] = make_uint4(L, L, g[0], g[1]); e4[1] = make_uint4(R, R, R, R);
        const unsigned E0 = ext[(s_ >> 2)], E1 = ext[(s_ >> 2) + 1];
        const unsigned A0 = __builtin_amdgcn_alignbyte(E1, E0, sh) ^ 0x80808080u, A1 = (E1 >> (8 * sh)) ^ 0x80808080u;
        h[r][0] = d_dot4k(A0, C0, 8192);
        h[r][1] = d_dot4(A1, C2, d_dot4k(A0, C1, 8192));
      }
    }
  } else if (sizeof(PIX) == 2) {
    const unsigned* tl = s_cl + (xf * 2 + (xa & 1)) * CL_STRIDE;
    unsigned U0[3], U1[3];
#pragma unroll
    for (int k = 0; k < 3; k++) { U0[k] = tl[k]; U1[k] = tl[3 + k]; }
    const M355_GLOBAL PIX* q = rp + (ptrdiff_t)(yi - 1) * rstride + (xa & ~1);
    unsigned S[7][3];
#pragma unroll
    for (int r = 0; r < 7; r++) { d_ldg12(q, S[r]); q += rstride; }
#pragma unroll
    for (int r = 0; r < 7; r++) {
      h[r][0] = d_dot2(S[r][2], U0[2], d_dot2(S[r][1], U0[1], d_dot2z(S[r][0], U0[0])));
      h[r][1] = d_dot2(S[r][2], U1[2], d_dot2(S[r][1], U1[1], d_dot2z(S[r][0], U1[0])));
    }
  } else {
    const unsigned* tl = s_cl + xf * 4;
    const unsigned C0 = tl[0], C1 = tl[1], C2 = tl[2];
    const unsigned sh = (unsigned)xa & 3u;
    const M355_GLOBAL PIX* q = rp + (ptrdiff_t)(yi - 1) * rstride + (xa & ~3);
    unsigned S[7][2];
#pragma unroll
    for (int r = 0; r < 7; r++) { d_ldg8(q, S[r]); q += rstride; }
#pragma unroll
    for (int r = 0; r < 7; r++) {
      const unsigned A0 = __builtin_amdgcn_alignbyte(S[r][1], S[r][0], sh) ^ 0x80808080u, A1 = (S[r][1] >> (8 * sh)) ^ 0x80808080u;
      h[r][0] = d_dot4k(A0, C0, 8192);
      h[r][1] = d_dot4(A1, C2, d_dot4k(A0, C1, 8192));
    }
  }
  h[7][0] = h[7][1] = 0;
  unsigned Q[4][2];
#pragma unroll
  for (int k = 0; k < 4; k++)
#pragma unroll
    for (int j = 0; j < 2; j++) Q[k][j] = sizeof(PIX) == 2 ? d_pack_mid16((unsigned)h[2 * k][j], (unsigned)h[2 * k + 1][j]) : d_pack_lo16((unsigned)h[2 * k][j], (unsigned)h[2 * k + 1][j]);
  const unsigned* ty = s_cv + yf * ET_STRIDE;
  unsigned YE[2], YO[3];
#pragma unroll
  for (int k = 0; k < 2; k++) YE[k] = ty[k];
#pragma unroll
  for (int k = 0; k < 3; k++) YO[k] = ty[2 + k];
#pragma unroll
  for (int m = 0; m < 2; m++) {
    int ve[2], vo[2];
#pragma unroll
    for (int j = 0; j < 2; j++) {
      ve[j] = d_dot2(Q[m + 1][j], YE[1], d_dot2z(Q[m][j], YE[0]));
      vo[j] = d_dot2(Q[m + 2][j], YO[2], d_dot2(Q[m + 1][j], YO[1], d_dot2z(Q[m][j], YO[0])));
    }
    pred[2 * m] = d_pack_mid16((unsigned)ve[0], (unsigned)ve[1]);
    pred[2 * m + 1] = d_pack_mid16((unsigned)vo[0], (unsigned)vo[1]);
  }
}

/* One job of the lean kernels (bit depths <= 12).  MODE 0: windows inside the picture, no explicit weights: the lists are the workgroup's
 * (bi: wave-uniform) and the write-back is packed 16-bit arithmetic; 1: explicit weights — one or two lists per lane, the 32-bit
 * write-back; 2: EDGE — windows that leave the picture (clamped rows), weights per lane. */
template <class PIX, int MODE>
__device__ __forceinline__ void d_inter_job_lean(const DevPic& p, uint32_t job, bool bi_u, const unsigned* s_ql, const unsigned* s_qv, const unsigned* s_cl, const unsigned* s_cv, unsigned* ext, const DevRef* s_refs);

/* one launch for the main and the edge job range: blocks [0, nblk_edge8) take edge jobs */
template <class PIX>
__global__ void __launch_bounds__(M355_INTER_BLOCK, M355_INTER_WAVES) k_inter_jobs(DevPic p)
{
  M355_GATE(p);
  /* the job counts live on the device (k_job_count / k_meta_pb); the grid is an upper bound: surplus workgroups leave here.
     Ranges: [0, tot0) one list, [tot0, tot1) two lists, [tot1, tot2) explicit weights, [tot2, tot3) EDGE (windows that leave the picture) */
  const int t0 = (int)p.job_tot[0], t1 = (int)p.job_tot[1], t2 = (int)p.job_tot[2], t3 = (int)p.job_tot[3];
  auto blocks8 = [](int jobs) { return (((jobs + M355_INTER_BLOCK - 1) / M355_INTER_BLOCK + 7) / 8) * 8; };
  const int nblk_uni8 = blocks8(t0), nblk_bi8 = blocks8(t1 - t0), nblk_w8 = blocks8(t2 - t1), nblk_edge8 = blocks8(t3 - t2);
  if ((int)blockIdx.x >= nblk_edge8 + nblk_bi8 + nblk_uni8 + nblk_w8) return;
  /* Dispatch order = expected cost, longest first: edge blocks (clamped loads make them latency-bound; started early they overlap with
     everything else; spread round-robin over the XCDs), then the other classes in an XCD-aware order: block b runs on XCD b % 8, and
     every XCD gets one contiguous eighth of each range (= compact regions of the picture) so reference-window overlap hits that XCD's
     own L2.  Within an XCD the classes are interleaved in proportion (all lists are in PB order, so block i/per_a of one and block
     i/per_b of another cover the same part of the picture): they then read the same reference region while it is still in that
     XCD's L2 — run one class after the other and every region is fetched twice, far apart in time. */
  const int b = blockIdx.x;
  int cls, ji, jend;
  if (b < nblk_edge8) { cls = 3; ji = t2 + b * M355_INTER_BLOCK; jend = t3; }
  else {
    const int bm = b - nblk_edge8, xcd = bm & 7, slot = bm >> 3;
    const int per_bi = nblk_bi8 >> 3, per_uni = nblk_uni8 >> 3, per_w = nblk_w8 >> 3;
    const int per = per_bi + per_uni + per_w;
    const int bi_before = (int)(((long long)slot * per_bi) / per), bi_after = (int)(((long long)(slot + 1) * per_bi) / per);
    if (bi_after != bi_before) { cls = 1; ji = t0 + (xcd * per_bi + bi_before) * M355_INTER_BLOCK; jend = t1; }
    else {
      const int s2 = slot - bi_before, per2 = per_uni + per_w;
      const int w_before = (int)(((long long)s2 * per_w) / per2), w_after = (int)(((long long)(s2 + 1) * per_w) / per2);
      if (w_after != w_before) { cls = 2; ji = t1 + (xcd * per_w + w_before) * M355_INTER_BLOCK; jend = t2; }
      else { cls = 0; ji = (xcd * per_uni + s2 - w_before) * M355_INTER_BLOCK; jend = t0; }
    }
  }
  if (ji >= jend) return;          /* (workgroup-uniform: the padding blocks of a range) */
  ji += threadIdx.x;

  /* the reference-frame table (plane pointers / pitches per DPB slot) in LDS: a job looks its references up
     with ds_reads instead of a dependent global load per list and component (each one a full memory
     latency on the wave's critical path) */
  __shared__ DevRef s_refs[M355_MAX_REF_FRAMES];
  {
    const unsigned* src = (const unsigned*)p.refs;
    unsigned* dst = (unsigned*)s_refs;
    for (int i = threadIdx.x; i < (int)(sizeof(s_refs) / 4); i += M355_INTER_BLOCK) dst[i] = src[i];
  }
  __shared__ __attribute__((aligned(16))) unsigned s_tab[LT_WORDS];
  if (threadIdx.x < LT_WORDS / 4) ((uint4*)s_tab)[threadIdx.x] = ((const uint4*)p.inter_tabs)[threadIdx.x];
  uint32_t job = 0;
  if (ji < jend) job = p.jobs[ji];          /* (requested beside the tables) */
  __syncthreads();
  if (ji >= jend) return;
  /* EDGE jobs extend their window rows in LDS: 20 words per lane (d_mc_luma_lean) */
  __shared__ __attribute__((aligned(16))) unsigned s_ext[M355_INTER_BLOCK * 20];
  unsigned* ext = s_ext + threadIdx.x * 20;
  if (cls == 3) d_inter_job_lean<PIX, 2>(p, job, false, s_tab + LT_QL, s_tab + LT_QV, s_tab + LT_CL, s_tab + LT_CV, ext, s_refs);
  else if (cls == 2) d_inter_job_lean<PIX, 1>(p, job, false, s_tab + LT_QL, s_tab + LT_QV, s_tab + LT_CL, s_tab + LT_CV, ext, s_refs);
  else d_inter_job_lean<PIX, 0>(p, job, cls == 1, s_tab + LT_QL, s_tab + LT_QV, s_tab + LT_CL, s_tab + LT_CV, ext, s_refs);
}

template <class PIX, int MODE>
__device__ __forceinline__ void d_inter_job_lean(const DevPic& p, uint32_t job, bool bi_u, const unsigned* s_ql, const unsigned* s_qv, const unsigned* s_cl, const unsigned* s_cv, unsigned* ext, const DevRef* s_refs)
{
  constexpr bool WEIGHTED = MODE != 0, EDGE = MODE == 2;
  const m355_pb pb = p.pbs[job & 0x1FFFFFFu];
  const int strip = (job >> 25) & 15, rblk = job >> 29;
  const int x0 = pb.x + 4 * strip, y0 = pb.y + 8 * rblk;
  const int rows = min(8, pb.h - 8 * rblk);          /* 4 or 8 */
  {
    uint32_t* po = p.pb_of + (size_t)(y0 >> 2) * p.w4 + (x0 >> 2);
    po[0] = (job & 0x1FFFFFFu) + 1;
    if (rows > 4) po[p.w4] = (job & 0x1FFFFFFu) + 1;
  }
  const bool mc0 = pb.flags & M355_PBF_MC_L0;
  const bool bi = WEIGHTED ? (mc0 && (pb.flags & M355_PBF_MC_L1)) : bi_u;
  const int npass = bi ? 2 : 1;
  const bool a1 = !mc0;
  const int refA = a1 ? pb.ref_slot[1] : pb.ref_slot[0], mvxA = a1 ? pb.mv[1][0] : pb.mv[0][0], mvyA = a1 ? pb.mv[1][1] : pb.mv[0][1];
  const bool fillA = pb.flags & (a1 ? M355_PBF_FILL_L1 : M355_PBF_FILL_L0), fillB = pb.flags & M355_PBF_FILL_L1;
  const int wtA = a1 ? pb.wt_idx[1] : pb.wt_idx[0], wtB = pb.wt_idx[1];
  const int nc = p.pp.chroma_format_idc ? 3 : 1;

  /* weights of component c as the one formula of d_wpred (WEIGHTED jobs only; see d_inter_job) */
  auto make_ws = [&](int c, int bd) {
    WtSel ws;
    const bool weighted = !EDGE || (pb.flags & M355_PBF_WEIGHTED) != 0;      /* (an edge job carries explicit weights or not, per lane) */
    m355_wt wa, wb;
    if (weighted) { wa = p.wts[wtA]; wb = p.wts[wtB]; }
    const int shift3 = max(2, 14 - bd), shift2 = max(3, 15 - bd);
    ws.w0 = 1; ws.w1 = bi ? 1 : 0; ws.o = 0;
    ws.sh = bi ? shift2 : shift3; ws.rnd = 1 << (ws.sh - 1);
    if (weighted) {
      const int o0 = c == 0 ? wa.o[0] : (c == 1 ? wa.o[1] : wa.o[2]), o1 = c == 0 ? wb.o[0] : (c == 1 ? wb.o[1] : wb.o[2]);
      const int log2WD = c ? wa.log2wd_chroma : wa.log2wd_luma;
      ws.w0 = c == 0 ? wa.w[0] : (c == 1 ? wa.w[1] : wa.w[2]);
      ws.w1 = bi ? (c == 0 ? wb.w[0] : (c == 1 ? wb.w[1] : wb.w[2])) : 0;
      ws.rnd = bi ? (int)((unsigned)(o0 + o1 + 1) << log2WD) : (1 << (log2WD - 1));
      ws.sh = bi ? log2WD + 1 : log2WD;
      ws.o = bi ? 0 : o0;
    }
    return ws;
  };
  /* packed write-back of one register pair (two samples) — put_unweighted_pred / put_weighted_pred_avg (fallback-motion.cc:33-84):
       one list : clip((a + rnd3) >> shift3),      shift3 = 14 - bd
       two lists: clip((a + b + rnd2) >> shift2),  shift2 = 15 - bd
     in SATURATING signed 16-bit arithmetic: a sum that leaves int16 is clipped by the reference as well (32767 >> shift2 is exactly the
     largest sample value, 32767 >> shift3 lies above it; -32768 >> s is negative), so the saturated sum gives the same sample. */
  /* two samples with per-lane weights (put_weighted_pred / _bipred, and the unweighted forms as weights 1 / 0: d_wpred's one formula):
     ((a w0 + b w1 + rnd) >> sh) + o = (a w0 + b w1 + rnd + (o << sh)) >> sh — one v_dot2 per sample on the pairs (a, b).  a, b: two
     samples each, packed */
  auto wt_pair = [&](unsigned a, unsigned b, unsigned wp, int rnd, int sh, int bd_) {
    const int lo = d_clip_bd(d_dot2(d_pack_lo16(a, b), wp, rnd) >> sh, bd_), hi = d_clip_bd(d_dot2(d_pack_hi16(a, b), wp, rnd) >> sh, bd_);
    return (unsigned)lo | ((unsigned)hi << 16);
  };
  auto pk_pred = [&](unsigned a, unsigned b, unsigned rnd, int sh, unsigned maxv) {
    unsigned t = bi ? d_pk_addsat_i16(a, b) : b;
    t = d_pk_ashr16(d_pk_addsat_i16(t, rnd), sh);
    t = d_pk_max_i16(t, 0u);
    return bi ? t : d_pk_min_i16(t, maxv);
  };

  /* ---- luma ---- */
  {
    const int bd = sizeof(PIX) == 1 ? 8 : p.pp.bit_depth_luma;
    unsigned pa[8][2];
#pragma unroll
    for (int y = 0; y < 8; y++) { pa[y][0] = 0; pa[y][1] = 0; }
#pragma unroll 1
    for (int pass = 0; pass < npass; pass++) {
      unsigned cur[8][2];
      if (pass ? fillB : fillA) {        /* reference missing: predSamples = 1 << 13 (motion.cc:362-376) */
#pragma unroll
        for (int y = 0; y < 8; y++) { cur[y][0] = 0x20002000u; cur[y][1] = 0x20002000u; }
      } else {
        const DevRef* ref = &s_refs[pass ? pb.ref_slot[1] : refA];
        const int mvx = pass ? pb.mv[1][0] : mvxA, mvy = pass ? pb.mv[1][1] : mvyA;
        d_mc_luma_lean<PIX, EDGE>((const M355_GLOBAL PIX*)ref->plane[0], ref->stride[0], p.pw[0], p.ph[0], x0 + (mvx >> 2), y0 + (mvy >> 2), mvx & 3, mvy & 3, s_ql, s_qv, ext, cur);
      }
      if (pass + 1 < npass) {            /* first list of a bi-predicted block: keep it for the second pass */
#pragma unroll
        for (int y = 0; y < 8; y++) { pa[y][0] = cur[y][0]; pa[y][1] = cur[y][1]; }
        continue;
      }
      M355_COMPILER_FENCE();          /* the weight loads must not be hoisted into the filter loops */
      PIX* d = (PIX*)p.plane[0] + (size_t)y0 * p.stride[0] + x0;
      if (WEIGHTED) {
        /* all four reference formulas as ONE dot2 per sample (see wt_pair): (a, b) . (w0, w1) + rnd', >> sh, clip */
        const WtSel ws = make_ws(0, bd);
        const unsigned wp = d_pack16(ws.w0, ws.w1);
        const int rnd = ws.rnd + (int)((unsigned)ws.o << ws.sh);
#pragma unroll
        for (int y = 0; y < 8; y++) {
          if (y >= rows) break;
          unsigned o[2];
#pragma unroll
          for (int jp = 0; jp < 2; jp++) o[jp] = wt_pair(bi ? pa[y][jp] : cur[y][jp], cur[y][jp], wp, rnd, ws.sh, bd);
          if (sizeof(PIX) == 2) d_st_nt8(d + (size_t)y * p.stride[0], o[0], o[1]);
          else d_st_nt4(d + (size_t)y * p.stride[0], d_pack_bytes(o[0], o[1]));
        }
      } else {
        const int sh = (bi ? 15 : 14) - bd;
        const unsigned rnd = (1u << (sh - 1)) * 0x10001u, maxv = ((1u << bd) - 1u) * 0x10001u;
#pragma unroll
        for (int y = 0; y < 8; y++) {
          if (y >= rows) break;
          unsigned o0 = pk_pred(pa[y][0], cur[y][0], rnd, sh, maxv), o1 = pk_pred(pa[y][1], cur[y][1], rnd, sh, maxv);
          /* streaming stores (k_asm.h): -4 % kernel time, -10 % fabric fetch (profiles/r02_b_inter_variants.txt) */
          if (sizeof(PIX) == 2) d_st_nt8(d + (size_t)y * p.stride[0], o0, o1);
          else d_st_nt4(d + (size_t)y * p.stride[0], d_pack_bytes(o0, o1));
        }
      }
    }
  }
  if (nc == 1) return;

  /* ---- chroma (4:2:0): 2 columns x 4 rows per plane, BOTH planes per pass so that their 14 window rows are in flight together;
     chroma mv = luma mv in 1/8 pel (motion.cc:196-203) ---- */
  {
    const int bd = sizeof(PIX) == 1 ? 8 : p.pp.bit_depth_chroma;
    const int xc = x0 >> 1, yc = y0 >> 1, crows = rows >> 1;
    unsigned pa1[4], pa2[4];
#pragma unroll
    for (int y = 0; y < 4; y++) { pa1[y] = 0; pa2[y] = 0; }
#pragma unroll 1
    for (int pass = 0; pass < npass; pass++) {
      unsigned cur1[4], cur2[4];
      if (pass ? fillB : fillA) {
#pragma unroll
        for (int y = 0; y < 4; y++) { cur1[y] = 0x20002000u; cur2[y] = 0x20002000u; }
      } else {
        const DevRef* ref = &s_refs[pass ? pb.ref_slot[1] : refA];
        const int mvx = pass ? pb.mv[1][0] : mvxA, mvy = pass ? pb.mv[1][1] : mvyA;
        d_mc_chroma_lean<PIX, EDGE>((const M355_GLOBAL PIX*)ref->plane[1], ref->stride[1], p.pw[1], p.ph[1], xc + (mvx >> 3), yc + (mvy >> 3), mvx & 7, mvy & 7, s_cl, s_cv, ext, cur1);
        d_mc_chroma_lean<PIX, EDGE>((const M355_GLOBAL PIX*)ref->plane[2], ref->stride[2], p.pw[2], p.ph[2], xc + (mvx >> 3), yc + (mvy >> 3), mvx & 7, mvy & 7, s_cl, s_cv, ext, cur2);
      }
      if (pass + 1 < npass) {
#pragma unroll
        for (int y = 0; y < 4; y++) { pa1[y] = cur1[y]; pa2[y] = cur2[y]; }
        continue;
      }
      M355_COMPILER_FENCE();
      PIX* d1 = (PIX*)p.plane[1] + (size_t)yc * p.stride[1] + xc;
      PIX* d2 = (PIX*)p.plane[2] + (size_t)yc * p.stride[2] + xc;
      if (WEIGHTED) {
        const WtSel ws1 = make_ws(1, bd), ws2 = make_ws(2, bd);
        const unsigned wp1 = d_pack16(ws1.w0, ws1.w1), wp2 = d_pack16(ws2.w0, ws2.w1);
        const int rnd1 = ws1.rnd + (int)((unsigned)ws1.o << ws1.sh), rnd2 = ws2.rnd + (int)((unsigned)ws2.o << ws2.sh);
#pragma unroll
        for (int y = 0; y < 4; y++) {
          if (y >= crows) break;
          const unsigned o1 = wt_pair(bi ? pa1[y] : cur1[y], cur1[y], wp1, rnd1, ws1.sh, bd);
          const unsigned o2 = wt_pair(bi ? pa2[y] : cur2[y], cur2[y], wp2, rnd2, ws2.sh, bd);
          if (sizeof(PIX) == 2) { d_st_nt4(d1 + (size_t)y * p.stride[1], o1); d_st_nt4(d2 + (size_t)y * p.stride[2], o2); }
          else {
            *(unsigned short*)(d1 + (size_t)y * p.stride[1]) = (unsigned short)d_pack_bytes(o1, 0u);
            *(unsigned short*)(d2 + (size_t)y * p.stride[2]) = (unsigned short)d_pack_bytes(o2, 0u);
          }
        }
      } else {
        const int sh = (bi ? 15 : 14) - bd;
        const unsigned rnd = (1u << (sh - 1)) * 0x10001u, maxv = ((1u << bd) - 1u) * 0x10001u;
#pragma unroll
        for (int y = 0; y < 4; y++) {
          if (y >= crows) break;
          unsigned o1 = pk_pred(pa1[y], cur1[y], rnd, sh, maxv), o2 = pk_pred(pa2[y], cur2[y], rnd, sh, maxv);
          if (sizeof(PIX) == 2) { d_st_nt4(d1 + (size_t)y * p.stride[1], o1); d_st_nt4(d2 + (size_t)y * p.stride[2], o2); }
          else {
            *(unsigned short*)(d1 + (size_t)y * p.stride[1]) = (unsigned short)d_pack_bytes(o1, 0u);
            *(unsigned short*)(d2 + (size_t)y * p.stride[2]) = (unsigned short)d_pack_bytes(o2, 0u);
          }
        }
      }
    }
  }
}

template <class PIX>
static void launch_jobs(const DevPic& p, hipStream_t st)
{
  /* each range's blocks are padded to a multiple of 8 for the XCD-contiguous block order; the counts are on the device, so the
     grid covers the most jobs the list can hold */
  const unsigned grid = (unsigned)((p.jobs_cap + M355_INTER_BLOCK - 1) / M355_INTER_BLOCK) + 4 * 8;
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k_inter_jobs<PIX>), dim3(grid), dim3(M355_INTER_BLOCK), 0, st, p);
}

void m355_launch_inter(const DevPic& p, bool hbd, hipStream_t st)
{
  if (!p.n_pbs) return;
  /* the job kernels: 4:2:0 / monochrome at bit depths up to 12 (every BASELINE configuration); everything else — 4:2:2 / 4:4:4, deeper planes,
     pictures narrower than one vector load — takes the one-wave-per-PB kernel, whose arithmetic is the reference's as written (m355_inter_uses_jobs, k_common.h) */
  if (m355_inter_uses_jobs(p)) {
    if (hbd) launch_jobs<uint16_t>(p, st); else launch_jobs<uint8_t>(p, st);
    return;
  }
  const dim3 grid((p.n_pbs + 3) / 4), block(256);
  if (hbd) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_inter_generic<uint16_t>), grid, block, 0, st, p);
  else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_inter_generic<uint8_t>), grid, block, 0, st, p);
}
