/*
 * k_residual.hip — inverse quantisation + inverse transform (DCT 4..32, DST 4x4, transform skip,
 * transquant bypass, RDPCM, rotation) for every coded transform block of a picture.
 *
 * Replaces scale_coefficients_internal (transform.cc:361-642) and the table slots it dispatches to:
 * dequant_coeff_block (fallback-dct.cc:1212-1220) + the int64 / scaling-list loops
 * (transform.cc:480-524), transform_idct_add / transform_4x4_luma_add (fallback-dct.cc:550-691,
 * 269-407), transform_skip_residual / rdpcm_* / transform_bypass* / rotate_coefficients
 * (fallback-dct.cc:81-256) and add_residual (fallback-dct.h:65-73).
 *
 * Mapping: blocks are binned by size on the host (m355_picture.rb_count).  A wavefront owns
 * 64/min(64,nT^2) blocks; the sparse (pos,level) pairs are dequantised and scattered into an LDS
 * tile, the separable transform runs column pass -> LDS -> row pass with the reference's exact
 * intermediate clip (int16 after the first stage, none after the second), pruned to the occupied
 * rows/columns (the reference's lastCol pruning, fallback-dct.cc:614-617, only skips zero terms).
 * Inter blocks are added to the picture in place; intra blocks go to the int16 residual buffer that
 * the intra wavefront consumes, so the serial intra chain never waits for a transform.
 * Roofline: HBM-bound (4 B per nonzero coefficient in, nT^2 samples read-modify-write); the matrix
 * work is VALU int MACs out of LDS, far below the integer roof at the sparsities of real streams.
 */
#include "k_common.h"

/* M[k][n] = c(k(2n+1)), c = quarter wave of the HEVC core transform (fallback-dct.cc:512-545) */
__constant__ int8_t c_dct_qw[33] = {64, 90, 90, 90, 89, 88, 87, 85, 83, 82, 80, 78, 75, 73, 70, 67, 64,
                                    61, 57, 54, 50, 46, 43, 38, 36, 31, 25, 22, 18, 13, 9,  4,  0};
__constant__ int8_t c_dst4[16] = {29, 55, 74, 84, 74, 74, 0, -74, 84, -29, -74, 55, 55, -84, 74, -29};
__constant__ int8_t c_level_scale[6] = {40, 45, 51, 57, 64, 72};

__device__ __forceinline__ int dct_wave(int m)
{
  m &= 127;
  if (m <= 32) return c_dct_qw[m];
  if (m <= 64) return -c_dct_qw[64 - m];
  if (m < 96) return -c_dct_qw[m - 64];
  return c_dct_qw[128 - m];
}

template <int LOG2, class PIX>
__global__ void __launch_bounds__(256) k_residual(DevPic p, int rb_base, int rb_n)
{
  constexpr int NT = 1 << LOG2, N2 = NT * NT;
  constexpr int G = N2 < 64 ? N2 : 64;   /* lanes per block */
  constexpr int TPW = 64 / G;            /* blocks per wave */
  constexpr int S = N2 / G;              /* samples per lane */
  constexpr int FACT = 32 / NT;

  __shared__ int8_t s_mat[32 * 32];
  __shared__ int16_t s_c[4 * TPW][N2];   /* dense coefficients */
  __shared__ int16_t s_g[4 * TPW][N2];   /* first-stage output */
  __shared__ int s_r[4 * TPW][N2];       /* residual of the skip / bypass paths (prefix sums need int32) */

  for (int i = threadIdx.x; i < 1024; i += 256) s_mat[i] = (int8_t)dct_wave((i >> 5) * (2 * (i & 31) + 1));
  __syncthreads();

  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int sub = lane / G, sl = lane % G;
  const int tbi = (blockIdx.x * 4 + wave) * TPW + sub;
  const bool active = tbi < rb_n;
  int16_t* cf = s_c[wave * TPW + sub];
  int16_t* gg = s_g[wave * TPW + sub];
  int* rr = s_r[wave * TPW + sub];

  m355_rb rb;
  if (active) rb = p.rbs[rb_base + tbi];
  else { rb.ncoeff = 0; rb.kind = M355_RK_DCT; rb.flags = 0; rb.cidx = 0; rb.qp = 0; rb.x = rb.y = 0; rb.coeff_ofs = rb.res_ofs = 0; rb.matrix_id = 0; rb.log2_size = LOG2; }
  const int bd = rb.cidx ? p.pp.bit_depth_chroma : p.pp.bit_depth_luma;

  /* ---- dequantise + scatter (transform.cc:408-525) ---- */
#pragma unroll
  for (int k = 0; k < S; k++) cf[sl + G * k] = 0;
  wave_sync();
  int maxrow = -1, maxcol = -1;
  {
    const bool raw = rb.kind == M355_RK_BYPASS || (rb.flags & M355_RBF_DEQUANTIZED);
    const bool sclist = (p.pp.flags & M355_PF_SCALING_LIST) != 0;
    int bdShift = bd + LOG2 - 5;
    if (!sclist) bdShift -= 4;
    const long long offset = 1ll << (bdShift - 1);
    const int ls = c_level_scale[rb.qp % 6], qs = rb.qp / 6;
    const uint8_t* scl = nullptr;
    if (sclist && p.scaling) {
      const int sz_ofs = LOG2 == 2 ? 0 : LOG2 == 3 ? 6 * 16 : LOG2 == 4 ? 6 * 16 + 6 * 64 : 6 * 16 + 6 * 64 + 6 * 256;
      scl = p.scaling + sz_ofs + rb.matrix_id * N2;
    }
    for (int k = sl; k < rb.ncoeff; k += G) {
      const uint32_t e = p.coeffs[rb.coeff_ofs + k];
      int pos = e & 0xFFFF;
      const int lvl = (int16_t)(e >> 16);
      if (pos >= N2) continue;
      int v;
      if (raw) v = lvl;
      else {
        const long long fact = (long long)(scl ? scl[pos] * ls : ls) << qs;
        long long t = ((long long)lvl * fact + offset) >> bdShift;
        v = t < -32768 ? -32768 : (t > 32767 ? 32767 : (int)t);
      }
      if (rb.flags & M355_RBF_ROTATE) pos = N2 - 1 - pos;
      cf[pos] = (int16_t)v;
      if (v != 0) { maxrow = max(maxrow, pos >> LOG2); maxcol = max(maxcol, pos & (NT - 1)); }
    }
  }
  /* reduce the occupied extent over the block's lanes */
#pragma unroll
  for (int m = G >> 1; m >= 1; m >>= 1) {
    maxrow = max(maxrow, __shfl_xor(maxrow, m, G));
    maxcol = max(maxcol, __shfl_xor(maxcol, m, G));
  }
  wave_sync();

  int res[S];
  if (rb.kind == M355_RK_DCT || rb.kind == M355_RK_DST) {
    const bool dst = (LOG2 == 2) && rb.kind == M355_RK_DST;
    /* column pass: g[i][c] = clip16((sum_j M[j][i] * coef[j][c] + 64) >> 7) */
#pragma unroll
    for (int k = 0; k < S; k++) {
      const int o = sl + G * k, c = o & (NT - 1), i = o >> LOG2;
      int sum = 0;
      if (c <= maxcol)
        for (int j = 0; j <= maxrow; j++) {
          const int m = dst ? c_dst4[j * 4 + i] : s_mat[(FACT * j) * 32 + i];
          sum += m * cf[c + j * NT];
        }
      gg[o] = (int16_t)d_clip3(-32768, 32767, (sum + 64) >> 7);
    }
    wave_sync();
    /* row pass: r[y][i] = (sum_j M[j][i] * g[y][j] + rnd) >> (20-bd), not clipped */
    const int postShift = 20 - bd, rnd2 = 1 << (postShift - 1);
#pragma unroll
    for (int k = 0; k < S; k++) {
      const int o = sl + G * k, i = o & (NT - 1), y = o >> LOG2;
      int sum = 0;
      for (int j = 0; j <= maxcol; j++) {
        const int m = dst ? c_dst4[j * 4 + i] : s_mat[(FACT * j) * 32 + i];
        sum += m * gg[y * NT + j];
      }
      res[k] = (sum + rnd2) >> postShift;
    }
  } else {
    /* transform skip / bypass, optional RDPCM (fallback-dct.cc:81-91, 161-225) */
    const bool skip = rb.kind == M355_RK_SKIP;
    const int bdShift2 = 20 - bd, tsShift = 5 + LOG2, rnd = skip ? (1 << (bdShift2 - 1)) : 0;
#pragma unroll
    for (int k = 0; k < S; k++) {
      const int o = sl + G * k;
      int c = cf[o];
      if (skip) c = ((int)((unsigned)c << tsShift) + rnd) >> bdShift2;
      rr[o] = c;
    }
    wave_sync();
    if (rb.flags & (M355_RBF_RDPCM_V | M355_RBF_RDPCM_H)) {
      if (sl < NT) {
        int sum = 0;
        if (rb.flags & M355_RBF_RDPCM_V) for (int y = 0; y < NT; y++) { sum += rr[y * NT + sl]; rr[y * NT + sl] = sum; }
        else for (int x = 0; x < NT; x++) { sum += rr[sl * NT + x]; rr[sl * NT + x] = sum; }
      }
      wave_sync();
    }
#pragma unroll
    for (int k = 0; k < S; k++) res[k] = rr[sl + G * k];
  }

  if (!active) return;
  if (rb.flags & M355_RBF_DEFERRED) {
    int16_t* out = p.resbuf + rb.res_ofs;
#pragma unroll
    for (int k = 0; k < S; k++) out[sl + G * k] = (int16_t)d_clip3(-32768, 32767, res[k]);
  } else {
    PIX* d = (PIX*)p.plane[rb.cidx];
    const int stride = p.stride[rb.cidx];
#pragma unroll
    for (int k = 0; k < S; k++) {
      const int o = sl + G * k, x = o & (NT - 1), y = o >> LOG2;
      PIX* q = d + (rb.y + y) * stride + rb.x + x;
      *q = (PIX)d_clip_bd((int)*q + res[k], bd);
    }
  }
}

template <class PIX>
static void launch_sizes(const DevPic& p, hipStream_t st)
{
  int base = 0;
  if (p.rb_count[0]) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_residual<2, PIX>), dim3((p.rb_count[0] + 15) / 16), dim3(256), 0, st, p, base, p.rb_count[0]);
  base += p.rb_count[0];
  if (p.rb_count[1]) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_residual<3, PIX>), dim3((p.rb_count[1] + 3) / 4), dim3(256), 0, st, p, base, p.rb_count[1]);
  base += p.rb_count[1];
  if (p.rb_count[2]) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_residual<4, PIX>), dim3((p.rb_count[2] + 3) / 4), dim3(256), 0, st, p, base, p.rb_count[2]);
  base += p.rb_count[2];
  if (p.rb_count[3]) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_residual<5, PIX>), dim3((p.rb_count[3] + 3) / 4), dim3(256), 0, st, p, base, p.rb_count[3]);
}

void m355_launch_residual(const DevPic& p, bool hbd, hipStream_t st)
{
  if (hbd) launch_sizes<uint16_t>(p, st);
  else launch_sizes<uint8_t>(p, st);
}
