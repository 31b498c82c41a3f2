/*
 * k_sao.hip — sample adaptive offset (band + 4 edge classes), deblocked picture -> output frame.
 *
 * Replaces apply_sample_adaptive_offset_sequential / apply_sao_internal (sao.cc:28-382).  The
 * reference filters in place from a copy of the deblocked plane; here the deblocked working planes
 * are the input and the DPB frame is the output, so every sample is read once and written once
 * (unfiltered samples are copied).  One thread per 4 adjacent samples (vector load/store), 256 samples of a row per wave; CTB
 * parameters are loaded once per thread.  Quirks reproduced: the CTB slice address used in the
 * slice-boundary test is looked up with COMPONENT coordinates (sao.cc:56), PCM / transquant-bypass
 * samples are skipped, picture-border and (when filtering across them is disabled) slice / tile
 * border neighbours suppress the edge offset.
 * Roofline: HBM-bound, 2*S*B bytes per CTB + 28 B of parameters.
 */
#include "k_common.h"

template <class PIX> struct Vec4;
template <> struct Vec4<uint8_t> { typedef uint32_t T; };
template <> struct Vec4<uint16_t> { typedef uint2 T; };

template <class PIX>
__global__ void __launch_bounds__(256) k_sao(DevPic p, int c)
{
  /* one thread = 4 horizontally adjacent samples (always inside one CTB: component CTB width >= 8,
     plane widths are multiples of 4); a wave covers 256 samples of one row */
  typedef typename Vec4<PIX>::T V4;
  const int x0 = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4, y = blockIdx.y * 4 + (threadIdx.x >> 6);
  const int width = p.pw[c], height = p.ph[c];
  if (x0 >= width || y >= height) return;
  const PIX* in = (const PIX*)p.plane[c];
  PIX* out = (PIX*)p.out_plane[c];
  const int is = p.stride[c], os = p.out_stride[c];
  union { V4 v; PIX s[4]; } cur, res;
  cur.v = *(const V4*)(in + y * is + x0);
  res.v = cur.v;

  const int csw = c ? (p.sw == 2) : 0, csh = c ? (p.sh == 2) : 0;
  const int l2w = p.pp.log2_ctb_size - csw, l2h = p.pp.log2_ctb_size - csh;
  const int xCtb = x0 >> l2w, yCtb = y >> l2h;
  const m355_ctb ctb = p.ctbs[yCtb * p.ctbW + xCtb];
  const m355_slice csl = p.slices[ctb.slice_idx];
  const bool enabled = c == 0 ? (csl.flags & M355_SF_SAO_LUMA) : (csl.flags & M355_SF_SAO_CHROMA);
  const int type = (ctb.sao_type >> (2 * c)) & 3;
  if (enabled && type != 0) {
    const int bd = c ? p.pp.bit_depth_chroma : p.pp.bit_depth_luma, maxv = (1 << bd) - 1;
    const bool plf = (p.pp.flags & M355_PF_PCM_LOOP_FILTER_DISABLE) != 0;
    const bool extended = ctb.flags & M355_CTBF_HAS_PCM_OR_BYPASS;
    const int yl = y << csh;
    if (type == 2) {
      const int cls = (ctb.sao_eo_class >> (2 * c)) & 3;
      int h0, h1, v0, v1;
      if (cls == 0) { h0 = -1; h1 = 1; v0 = 0; v1 = 0; }
      else if (cls == 1) { h0 = 0; h1 = 0; v0 = -1; v1 = 1; }
      else if (cls == 2) { h0 = -1; h1 = 1; v0 = -1; v1 = 1; }
      else { h0 = 1; h1 = -1; v0 = -1; v1 = 1; }
      const int xC = xCtb << l2w, yC = yCtb << l2h;
      const int nSW = 1 << l2w, nSH = 1 << l2h;
      const int ctbW_ = (xC + nSW > width) ? width - xC : nSW, ctbH_ = (yC + nSH > height) ? height - yC : nSH;
      const int j = y - yC;
      /* neighbour rows, 6 samples each (x0-1 .. x0+4), only where they exist */
      int ra[6], rb[6];
      const int ya = y + v0, yb = y + v1;
#pragma unroll
      for (int t = 0; t < 6; t++) {
        const int xx = x0 - 1 + t;
        const bool okx = xx >= 0 && xx < width;
        ra[t] = (okx && ya >= 0 && ya < height) ? in[ya * is + xx] : 0;
        rb[t] = (okx && yb >= 0 && yb < height) ? in[yb * is + xx] : 0;
      }
      /* sao.cc:56 — component coordinates used as luma coordinates */
      const int ctbSliceAddrRS = d_slice_at(p, min(xC, p.pp.width - 1), min(yC, p.pp.height - 1)).slice_addr_rs;
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
        if (i == 0 || j == 0 || i == ctbW_ - 1 || j == ctbH_ - 1) {
#pragma unroll
          for (int k = 0; k < 2; k++) {
            const int xS = x + (k ? h1 : h0), yS = y + (k ? v1 : v0);
            if (xS < 0 || yS < 0 || xS >= width || yS >= height) { zero = true; break; }
            const m355_slice shN = d_slice_at(p, xS << csw, yS << csh);
            if (shN.slice_addr_rs < ctbSliceAddrRS && !(d_slice_at(p, xl, yl).flags & M355_SF_LF_ACROSS_SLICES)) { zero = true; break; }
            if (shN.slice_addr_rs > ctbSliceAddrRS && !(shN.flags & M355_SF_LF_ACROSS_SLICES)) { zero = true; break; }
            if (!(p.pp.flags & M355_PF_LF_ACROSS_TILES) &&
                p.tile_id[(xS >> l2w) + (yS >> l2h) * p.ctbW] != p.tile_id[(xC >> l2w) + (yC >> l2h) * p.ctbW]) { zero = true; break; }
          }
        }
        if (zero) continue;
        const int cv = cur.s[s];
        const int a = ra[s + 1 + h0], b = rb[s + 1 + h1];
        const int edgeIdx = d_sign(cv - a) + d_sign(cv - b);
        int off = 0;
        if (edgeIdx == -2) off = ctb.sao_offset[c][0];
        else if (edgeIdx == -1) off = ctb.sao_offset[c][1];
        else if (edgeIdx == 1) off = ctb.sao_offset[c][2];
        else if (edgeIdx == 2) off = ctb.sao_offset[c][3];
        res.s[s] = (PIX)d_clip3(0, maxv, cv + off);
      }
    } else {
#pragma unroll
      for (int s = 0; s < 4; s++) {
        if (extended) {
          const uint32_t ci = d_cu_index_at(p, (x0 + s) << csw, yl);
          if (ci) {
            const m355_cu cu = p.cus[ci - 1];
            if ((plf && (cu.flags & M355_CUF_PCM)) || (cu.flags & M355_CUF_TRANSQUANT_BYPASS)) continue;
          }
        }
        const int cv = cur.s[s];
        const int band = d_clip3(0, maxv, cv) >> (bd - 5);
        const int k = (band - ctb.sao_band_pos[c]) & 31;
        if (k < 4) res.s[s] = (PIX)d_clip3(0, maxv, cv + ctb.sao_offset[c][k]);
      }
    }
  }
  *(V4*)(out + y * os + x0) = res.v;
}

void m355_launch_sao(const DevPic& p, bool hbd, hipStream_t st)
{
  const int nc = p.pp.chroma_format_idc ? 3 : 1;
  for (int c = 0; c < nc; c++) {
    const dim3 grid((p.pw[c] + 255) / 256, (p.ph[c] + 3) / 4), block(256);
    if (hbd) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_sao<uint16_t>), grid, block, 0, st, p, c);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_sao<uint8_t>), grid, block, 0, st, p, c);
  }
}
