/*
 * k_inter.hip — motion-compensated inter prediction for all prediction blocks of a picture.
 *
 * Replaces, per PB, the whole of generate_inter_prediction_samples (motion.cc:288-730):
 *   mc_luma / mc_chroma incl. the picture-edge coordinate clamp (motion.cc:48-282),
 *   put_hevc_qpel / put_hevc_epel* (fallback-motion.cc:492-636, 305-415, 431-485, 262-302),
 *   put_unweighted_pred / put_weighted_pred_avg / put_weighted_pred / put_weighted_bipred
 *   (fallback-motion.cc:33-256).
 *
 * Mapping: one 64-lane wavefront per prediction block, walking the block in 16x16 (per component)
 * tiles.  Per tile and list the wave stages the (tw+7)x(th+7) reference window in LDS with clamped,
 * row-coalesced loads, runs the horizontal taps into an int16 LDS tile (the reference's mcbuffer,
 * same truncation), then the vertical taps into registers, and finally the weighted combination
 * straight to the picture — the 14-bit intermediates never touch HBM.  No workgroup barrier: the
 * four waves of a workgroup are independent (wave-level LDS ordering only).
 * Roofline: HBM/L2-bound; algorithmic bytes per PB and list = [(w+7)(h+7)+2(w/2+3)(h/2+3)]*B read,
 * w*h*1.5*B written.
 */
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
__global__ void __launch_bounds__(256) k_inter(DevPic p)
{
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

void m355_launch_inter(const DevPic& p, bool hbd, hipStream_t st)
{
  if (!p.n_pbs) return;
  const dim3 grid((p.n_pbs + 3) / 4), block(256);
  if (hbd) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_inter<uint16_t>), grid, block, 0, st, p);
  else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_inter<uint8_t>), grid, block, 0, st, p);
}
