/*
 * k_sao.hip — sample adaptive offset (band + 4 edge classes), deblocked picture -> output frame.
 *
 * Replaces apply_sample_adaptive_offset_sequential / apply_sao_internal (sao.cc:28-382).  The
 * reference filters in place from a copy of the deblocked plane; here the deblocked working planes
 * are the input and the DPB frame is the output, so every sample is read once and written once
 * (unfiltered samples are copied).  One thread per 4x4 sample block: six aligned 4-sample row vectors
 * plus cross-lane shuffles give the whole 6x6 neighbourhood, so the L2 sees 1.5 reads per sample instead
 * of 3 vector + 12 scalar loads per 4 samples; all components in one launch (grid.z); CTB parameters are
 * loaded once per thread.  Quirks reproduced: the CTB slice address used in the
 * slice-boundary test is looked up with COMPONENT coordinates (sao.cc:56), PCM / transquant-bypass
 * samples are skipped, picture-border and (when filtering across them is disabled) slice / tile
 * border neighbours suppress the edge offset.
 * Roofline: HBM-bound, 2*S*B bytes per CTB + 28 B of parameters.
 */
#include "k_common.h"

template <class PIX> struct Vec4;
template <> struct Vec4<uint8_t> { typedef uint32_t T; };
template <> struct Vec4<uint16_t> { typedef uint2 T; };

/* offsets of the two neighbours per edge class (sao.cc:83-88) */
template <int H0, int V0, class PIX>
__device__ __forceinline__ void d_sao_edge_block(const DevPic& p, int o0, int o1, int o2, int o3, const int nb[6][6], bool extended, bool plf, int x0, int y0,
                                                 int xC, int yC, int ctbW_, int ctbH_, int width, int height, int csw, int csh, int l2w, int l2h,
                                                 uint32_t nbmask, int maxv, int rows, PIX res[4][4])
{
  /* nb[r][k]: sample at (x0 - 1 + k, y0 - 1 + r); class offsets: a = (+H0,+V0), b = (-H0,-V0) */
#pragma unroll
  for (int j = 0; j < 4; j++) {
    if (j >= rows) break;
    const int y = y0 + j, jj = y - yC, yl = y << csh;
#pragma unroll
    for (int s = 0; s < 4; s++) {
      const int x = x0 + s, i = x - xC, xl = x << csw;
      if (extended) {
        const uint32_t ci = d_cu_index_at(p, xl, yl);
        if (ci) {
          const m355_cu cu = p.cus[ci - 1];
          if ((plf && (cu.flags & M355_CUF_PCM)) || (cu.flags & M355_CUF_TRANSQUANT_BYPASS)) continue;
        }
      }
      bool zero = false;
      if (i == 0 || jj == 0 || i == ctbW_ - 1 || jj == ctbH_ - 1) {
        /* CTB-border sample (sao.cc:122-164): both neighbours must lie in usable CTBs (k_meta_sao mask) */
#pragma unroll
        for (int k = 0; k < 2; k++) {
          const int xS = x + (k ? -H0 : H0), yS = y + (k ? -V0 : V0);
          const int dx = (xS >> l2w) - (xC >> l2w), dy = (yS >> l2h) - (yC >> l2h);
          if (xS < 0 || yS < 0 || xS >= width || yS >= height || ((nbmask >> ((dy + 1) * 3 + dx + 1)) & 1u)) zero = true;
        }
      }
      if (zero) continue;
      const int cv = nb[1 + j][1 + s];
      const int a = nb[1 + j + V0][1 + s + H0], b = nb[1 + j - V0][1 + s - H0];
      const int edgeIdx = d_sign(cv - a) + d_sign(cv - b);
      /* offsets in the order of sao.cc:95-100: edgeIdx -2,-1,+1,+2 -> saoOffsetVal 0..3 */
      const int off = edgeIdx == -2 ? o0 : (edgeIdx == -1 ? o1 : (edgeIdx == 1 ? o2 : (edgeIdx == 2 ? o3 : 0)));
      res[j][s] = (PIX)d_clip3(0, maxv, cv + off);
    }
  }
}

template <class PIX>
__global__ void __launch_bounds__(256) k_sao(DevPic p)
{
  /* one thread = a 4x4 sample block (always inside one CTB: component CTB sizes are >= 8, plane sizes
     multiples of 4... heights multiples of 2: the row count is clipped); a wave covers 256 x 4 samples.
     Rows y0-1 .. y0+4 are loaded once as aligned 4-sample vectors; the columns x0-1 and x0+4 come from
     the neighbouring lanes' vectors (cross-lane shuffle), from memory only at the wave's two ends. */
  typedef typename Vec4<PIX>::T V4;
  const int c = blockIdx.z;
  const int lane = threadIdx.x & 63;
  const int x0 = (blockIdx.x * 64 + lane) * 4, y0 = (blockIdx.y * 4 + (threadIdx.x >> 6)) * 4;
  const int width = p.pw[c], height = p.ph[c];
  if (y0 >= height || (int)blockIdx.x * 256 >= width) return;   /* wave-uniform (chroma planes are smaller than the grid) */
  const bool valid = x0 < width;
  const int rows = min(4, height - y0);
  const PIX* in = (const PIX*)p.plane[c];
  PIX* out = (PIX*)p.out_plane[c];
  const int is = p.stride[c], os = p.out_stride[c];

  const int csw = c ? (p.sw == 2) : 0, csh = c ? (p.sh == 2) : 0;
  const int l2w = p.pp.log2_ctb_size - csw, l2h = p.pp.log2_ctb_size - csh;
  const int xq = valid ? x0 : 0;
  const int xCtb = xq >> l2w, yCtb = y0 >> l2h;
  const m355_ctb ctb = p.ctbs[yCtb * p.ctbW + xCtb];
  const m355_slice csl = p.slices[ctb.slice_idx];
  /* this component's parameters, selected without indexing the record dynamically */
  const int band_pos = c == 0 ? ctb.sao_band_pos[0] : (c == 1 ? ctb.sao_band_pos[1] : ctb.sao_band_pos[2]);
  const int so0 = c == 0 ? ctb.sao_offset[0][0] : (c == 1 ? ctb.sao_offset[1][0] : ctb.sao_offset[2][0]);
  const int so1 = c == 0 ? ctb.sao_offset[0][1] : (c == 1 ? ctb.sao_offset[1][1] : ctb.sao_offset[2][1]);
  const int so2 = c == 0 ? ctb.sao_offset[0][2] : (c == 1 ? ctb.sao_offset[1][2] : ctb.sao_offset[2][2]);
  const int so3 = c == 0 ? ctb.sao_offset[0][3] : (c == 1 ? ctb.sao_offset[1][3] : ctb.sao_offset[2][3]);
  const bool enabled = c == 0 ? (csl.flags & M355_SF_SAO_LUMA) : (csl.flags & M355_SF_SAO_CHROMA);
  const int type = (enabled && valid) ? ((ctb.sao_type >> (2 * c)) & 3) : 0;
  const bool edge = type == 2;

  /* ---- load the 6x6 neighbourhood ---- */
  int nb[6][6];
  union { V4 v; PIX s[4]; } row[6];
  const bool any_edge = __any(edge);               /* a neighbour lane may need this lane's rows y0-1 / y0+4 */
#pragma unroll
  for (int r = 0; r < 6; r++) {
    const int yy = y0 - 1 + r;
    const bool need = valid && yy >= 0 && yy < height && (any_edge || (r >= 1 && r <= 4));
    if (need) row[r].v = *(const V4*)(in + (size_t)yy * is + x0);
    else { for (int k = 0; k < 4; k++) row[r].s[k] = 0; }
  }
#pragma unroll
  for (int r = 0; r < 6; r++) {
#pragma unroll
    for (int k = 0; k < 4; k++) nb[r][1 + k] = row[r].s[k];
    nb[r][0] = nb[r][5] = 0;
    if (any_edge) {
      const int yy = y0 - 1 + r;
      const bool yok = yy >= 0 && yy < height;
      int l = __shfl_up((int)row[r].s[3], 1, 64), rg = __shfl_down((int)row[r].s[0], 1, 64);
      if (edge && yok) {
        if (lane == 0 && x0 > 0) l = in[(size_t)yy * is + x0 - 1];
        if (lane == 63 && x0 + 4 < width) rg = in[(size_t)yy * is + x0 + 4];
      }
      nb[r][0] = l; nb[r][5] = rg;
    }
  }
  if (!valid) return;

  PIX res[4][4];
#pragma unroll
  for (int j = 0; j < 4; j++)
#pragma unroll
    for (int s2 = 0; s2 < 4; s2++) res[j][s2] = row[1 + j].s[s2];

  if (type != 0) {
    const int bd = c ? p.pp.bit_depth_chroma : p.pp.bit_depth_luma, maxv = (1 << bd) - 1;
    const bool plf = (p.pp.flags & M355_PF_PCM_LOOP_FILTER_DISABLE) != 0;
    const bool extended = ctb.flags & M355_CTBF_HAS_PCM_OR_BYPASS;
    if (edge) {
      const int cls = (ctb.sao_eo_class >> (2 * c)) & 3;
      const int xC = xCtb << l2w, yC = yCtb << l2h;
      const int nSW = 1 << l2w, nSH = 1 << l2h;
      const int ctbW_ = (xC + nSW > width) ? width - xC : nSW, ctbH_ = (yC + nSH > height) ? height - yC : nSH;
      const uint32_t nbmask = p.sao_nb[c * p.nCtb + yCtb * p.ctbW + xCtb];
      /* class -> first neighbour offset (sao.cc:83-88): 0:(-1,0) 1:(0,-1) 2:(-1,-1) 3:(+1,-1); the second is its mirror */
      if (cls == 0) d_sao_edge_block<-1, 0, PIX>(p, so0, so1, so2, so3, nb, extended, plf, x0, y0, xC, yC, ctbW_, ctbH_, width, height, csw, csh, l2w, l2h, nbmask, maxv, rows, res);
      else if (cls == 1) d_sao_edge_block<0, -1, PIX>(p, so0, so1, so2, so3, nb, extended, plf, x0, y0, xC, yC, ctbW_, ctbH_, width, height, csw, csh, l2w, l2h, nbmask, maxv, rows, res);
      else if (cls == 2) d_sao_edge_block<-1, -1, PIX>(p, so0, so1, so2, so3, nb, extended, plf, x0, y0, xC, yC, ctbW_, ctbH_, width, height, csw, csh, l2w, l2h, nbmask, maxv, rows, res);
      else d_sao_edge_block<1, -1, PIX>(p, so0, so1, so2, so3, nb, extended, plf, x0, y0, xC, yC, ctbW_, ctbH_, width, height, csw, csh, l2w, l2h, nbmask, maxv, rows, res);
    } else {
#pragma unroll
      for (int j = 0; j < 4; j++) {
        if (j >= rows) break;
#pragma unroll
        for (int s2 = 0; s2 < 4; s2++) {
          if (extended) {
            const uint32_t ci = d_cu_index_at(p, (x0 + s2) << csw, (y0 + j) << csh);
            if (ci) {
              const m355_cu cu = p.cus[ci - 1];
              if ((plf && (cu.flags & M355_CUF_PCM)) || (cu.flags & M355_CUF_TRANSQUANT_BYPASS)) continue;
            }
          }
          const int cv = nb[1 + j][1 + s2];
          const int band = d_clip3(0, maxv, cv) >> (bd - 5);
          const int k = (band - band_pos) & 31;
          if (k < 4) res[j][s2] = (PIX)d_clip3(0, maxv, cv + (k == 0 ? so0 : (k == 1 ? so1 : (k == 2 ? so2 : so3))));
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 4; j++) {
    if (j >= rows) break;
    union { V4 v; PIX s[4]; } o;
#pragma unroll
    for (int s2 = 0; s2 < 4; s2++) o.s[s2] = res[j][s2];
    *(V4*)(out + (size_t)(y0 + j) * os + x0) = o.v;
  }
}

void m355_launch_sao(const DevPic& p, bool hbd, hipStream_t st)
{
  /* one launch for all components: grid.z = component; chroma blocks beyond the chroma plane exit at once */
  const int nc = p.pp.chroma_format_idc ? 3 : 1;
  const dim3 grid((p.pw[0] + 255) / 256, (p.ph[0] + 15) / 16, nc), block(256);
  if (hbd) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_sao<uint16_t>), grid, block, 0, st, p);
  else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_sao<uint8_t>), grid, block, 0, st, p);
}
