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
#include <algorithm>
#include "k_common.h"

template <class PIX> struct Vec4;
template <> struct Vec4<uint8_t> { typedef uint32_t T; };
template <> struct Vec4<uint16_t> { typedef uint2 T; };

/* four horizontally adjacent samples as raw words / one sample out of them */
template <class PIX> __device__ __forceinline__ void d_sao_load4(const PIX* q, uint32_t* w)
{
  if (sizeof(PIX) == 2) { const uint2 v = *(const uint2*)q; w[0] = v.x; w[1] = v.y; }
  else w[0] = *(const uint32_t*)q;
}
template <class PIX> __device__ __forceinline__ int d_sao_sample(const uint32_t* w, int k)
{
  if (sizeof(PIX) == 2) return (int)((w[k >> 1] >> (16 * (k & 1))) & 0xFFFFu);
  return (int)((w[0] >> (8 * k)) & 0xFFu);
}

/* Edge offset for one 4x4 block, class given by the first neighbour offset (H0,V0) (sao.cc:83-88; the
 * second neighbour is its mirror).  Branch-free per sample: which samples are on the CTB ring and which
 * neighbour CTB a ring sample looks into are compile-time functions of (s,j,H0,V0) combined with the
 * thread's four position flags; availability is one bit of the k_meta_sao mask; skipped samples
 * (pcm/bypass, sao.cc:103-120) arrive as a 16-bit per-sample mask.
 * nb[r][k]: sample at (x0 - 1 + k, y0 - 1 + r). */
template <int H0, int V0, class PIX>
__device__ __forceinline__ void d_sao_edge_block(int o0, int o1, int o2, int o3, const int nb[6][6], uint32_t skipmask, bool L, bool R, bool T, bool Bt,
                                                 uint32_t nbmask, int maxv, int rows, PIX res[4][4])
{
  const unsigned tab_lo = ((unsigned)o0 & 0xFFu) | (((unsigned)o1 & 0xFFu) << 8) | (((unsigned)o2 & 0xFFu) << 24), tab_hi = (unsigned)o3 & 0xFFu;
#pragma unroll
  for (int j = 0; j < 4; j++) {
#pragma unroll
    for (int s = 0; s < 4; s++) {
      const bool lastrow = j == rows - 1;
      /* on the CTB ring? (the reference tests availability only there, sao.cc:122) */
      const bool ring = (s == 0 && L) || (s == 3 && R) || (j == 0 && T) || (lastrow && Bt);
      bool blocked = false;
#pragma unroll
      for (int k = 0; k < 2; k++) {
        const int sx = s + (k ? -H0 : H0), sy = j + (k ? -V0 : V0);
        const int dx = sx < 0 ? (L ? -1 : 0) : (sx > 3 ? (R ? 1 : 0) : 0);
        const int dy = sy < 0 ? (T ? -1 : 0) : ((sy > 3 || (V0 != 0 && lastrow && sy > j)) ? (Bt ? 1 : 0) : 0);
        blocked = blocked || ((nbmask >> ((dy + 1) * 3 + dx + 1)) & 1u);
      }
      const bool apply = !(ring && blocked) && !((skipmask >> (j * 4 + s)) & 1u);
      const int cv = nb[1 + j][1 + s];
      const int a = nb[1 + j + V0][1 + s + H0], b = nb[1 + j - V0][1 + s - H0];
      /* edgeIdx = Sign(c-a) + Sign(c-b); Sign(x) = med3(x,-1,1); offsets in the order of sao.cc:95-100 */
      const int edgeIdx = d_clip3(-1, 1, cv - a) + d_clip3(-1, 1, cv - b);
      const int off = (int)(int8_t)d_byte_lookup(tab_hi, tab_lo, (unsigned)(edgeIdx + 2));   /* [o0, o1, 0, o2, o3] */
      const int m = -(int)apply;
      res[j][s] = (PIX)((d_clip3(0, maxv, cv + off) & m) | (cv & ~m));
    }
  }
}

/* Packed path (bit depth <= 15): which of the 16 samples of a 4x4 block (bit j*4+s) sit on the CTB ring AND have one of
 * their two edge-class neighbours (direction +-(H0,V0)) in a CTB that may not be used (sao.cc:122-164)?  The neighbour of
 * sample (s,j) in direction (h,v) leaves the block in x iff s is on the matching block edge — the sample sets are
 * compile-time masks; which neighbouring CTB that means depends on whether the block edge is also a CTB edge (L,R,T,Bt). */
template <int h, int v>
__device__ __forceinline__ uint32_t d_sao_dir_blocked(bool L, bool R, bool T, bool Bt, uint32_t nbmask)
{
  constexpr uint32_t MX = h < 0 ? 0x1111u : (h > 0 ? 0x8888u : 0u), MY = v < 0 ? 0x000Fu : (v > 0 ? 0xF000u : 0u);
  constexpr uint32_t M_xy = MX & MY, M_x = MX & ~MY, M_y = MY & ~MX, M_in = 0xFFFFu & ~(MX | MY);
  const int dx = h < 0 ? -(int)L : (h > 0 ? (int)R : 0), dy = v < 0 ? -(int)T : (v > 0 ? (int)Bt : 0);
  const uint32_t b_in = (nbmask >> 4) & 1u, b_x = (nbmask >> (4 + dx)) & 1u, b_y = (nbmask >> (4 + 3 * dy)) & 1u, b_xy = (nbmask >> (4 + 3 * dy + dx)) & 1u;
  return (b_in ? M_in : 0u) | (b_x ? M_x : 0u) | (b_y ? M_y : 0u) | (b_xy ? M_xy : 0u);
}
template <int H0, int V0>
__device__ __forceinline__ uint32_t d_sao_bad(bool L, bool R, bool T, bool Bt, uint32_t nbmask)
{
  const uint32_t ring = (L ? 0x1111u : 0u) | (R ? 0x8888u : 0u) | (T ? 0x000Fu : 0u) | (Bt ? 0xF000u : 0u);
  return ring & (d_sao_dir_blocked<H0, V0>(L, R, T, Bt, nbmask) | d_sao_dir_blocked<-H0, -V0>(L, R, T, Bt, nbmask));
}
/* table index (edgeIdx + 2, sao.cc:95-100) of two packed samples: Sign(c-a) + Sign(c-b) + 2 on 16-bit lanes */
__device__ __forceinline__ uint32_t d_sao_edge_idx(uint32_t c, uint32_t a, uint32_t b)
{
  const uint32_t one = 0x00010001u, m1 = 0xFFFFFFFFu;
  const uint32_t s1 = d_pk_max_i16(d_pk_min_i16(d_pk_sub16(c, a), one), m1), s2 = d_pk_max_i16(d_pk_min_i16(d_pk_sub16(c, b), one), m1);
  return d_pk_add16(d_pk_add16(s1, s2), 0x00020002u);
}

template <class PIX, bool PACKED>
__device__ __forceinline__ void k_sao_body(const DevPic& p, const int c, const int bx, const int by)
{
  M355_GATE(p);
  /* one thread = a 4x4 sample block (always inside one CTB: component CTB sizes are >= 8 and planes are
     multiples of 4 wide; the row count is clipped at the picture bottom).  A wave covers 64 x 16 samples
     = 16 x 4 threads, i.e. it stays inside ONE 64x64 luma CTB, so the SAO type / edge class branches
     are wave-uniform.  Every thread loads only its own four rows as aligned 4-sample vectors; the rows
     above/below come from the lanes 16 up/down and the columns left/right from the neighbouring lanes
     (cross-lane shuffles) — memory is touched again only on the wave tile's rim. */
  typedef typename Vec4<PIX>::T V4;
  const int lane = threadIdx.x & 63, lx = lane & 15, ly = lane >> 4;
  const int width = p.pw[c], height = p.ph[c];
  const int xt = (bx * 4 + (int)(threadIdx.x >> 6)) * 64, yt = by * 16;
  if (yt >= height || xt >= width) return;        /* wave-uniform (chroma planes are smaller than the grid) */
  const int x0 = xt + lx * 4, y0 = yt + ly * 4;
  const bool valid = x0 < width && y0 < height;
  const int rows = valid ? min(4, height - y0) : 0;
  const PIX* in = (const PIX*)p.plane[c];
  PIX* out = (PIX*)p.out_plane[c];
  const int is = p.stride[c], os = p.out_stride[c];

  const int csw = c ? (p.sw == 2) : 0, csh = c ? (p.sh == 2) : 0;
  const int l2w = p.pp.log2_ctb_size - csw, l2h = p.pp.log2_ctb_size - csh;
  const int xCtb = (valid ? x0 : xt) >> l2w, yCtb = (valid ? y0 : yt) >> l2h;
  /* tile sharding: only own CTBs are filtered / written; waves without any own sample leave at once */
  const bool owned = !p.ctb_owner || p.ctb_owner[yCtb * p.ctbW + xCtb] != 0;
  if (!__any(owned)) return;

  /* ---- memory round trip 1: the CTB record, its neighbour mask and the thread's own four rows, all requested together and
     from CLAMPED addresses (no per-lane branch in front of a load: a branch makes hipcc wait for everything in flight — the
     kernel used to be a chain of up to 17 dependent round trips per wave, and it is latency x occupancy that bounds it, not
     bandwidth) ---- */
  constexpr int NW = (int)sizeof(V4) / 4;
  const int xs = max(0, min(x0, width - 4));                      /* (planes are at least 4 wide) */
  const m355_ctb ctb = p.ctbs[yCtb * p.ctbW + xCtb];
  const uint32_t nbmask_raw = p.sao_nb[c * p.nCtb + yCtb * p.ctbW + xCtb];
  uint32_t rw[6][NW];
#pragma unroll
  for (int r = 1; r <= 4; r++) d_sao_load4<PIX>(in + (size_t)min(y0 + r - 1, height - 1) * is + xs, rw[r]);
  /* this component's parameters, selected without indexing the record dynamically */
  const int band_pos = c == 0 ? ctb.sao_band_pos[0] : (c == 1 ? ctb.sao_band_pos[1] : ctb.sao_band_pos[2]);
  const int so0 = c == 0 ? ctb.sao_offset[0][0] : (c == 1 ? ctb.sao_offset[1][0] : ctb.sao_offset[2][0]);
  const int so1 = c == 0 ? ctb.sao_offset[0][1] : (c == 1 ? ctb.sao_offset[1][1] : ctb.sao_offset[2][1]);
  const int so2 = c == 0 ? ctb.sao_offset[0][2] : (c == 1 ? ctb.sao_offset[1][2] : ctb.sao_offset[2][2]);
  const int so3 = c == 0 ? ctb.sao_offset[0][3] : (c == 1 ? ctb.sao_offset[1][3] : ctb.sao_offset[2][3]);
  const int type_raw = valid ? ((ctb.sao_type >> (2 * c)) & 3) : 0;
  /* ---- (still round trip 1) the wave tile's rim: the row above the top lanes and below the bottom lanes, and the sample left / right
     of the outer lanes for the six rows — requested whether or not a CTB of the wave asks for an edge class: the kernel is bound by
     dependent round trips per wave at full occupancy, not by bytes (profiles/r04_af_*: 31 % less fabric fetch moved nothing), and the
     rim behind the CTB record was a second trip.  The slice's SAO switch comes with the neighbour mask (k_meta_sao, bit 15): no slice
     record either.  The rim columns are 48 samples per wave: lane i fetches sample i (ONE load instruction), the outer lanes pick
     theirs up with cross-lane reads. ---- */
  uint32_t rim_up[NW], rim_dn[NW];
  uint32_t rim_col = 0;
  {
    d_sao_load4<PIX>(in + (size_t)(ly == 0 ? max(y0 - 1, 0) : min(y0, height - 1)) * is + xs, rim_up);
    d_sao_load4<PIX>(in + (size_t)(ly == 3 ? min(y0 + 4, height - 1) : min(y0 + 3, height - 1)) * is + xs, rim_dn);
    /* lane i < 48: group g = i / 12 (the lanes with ly == g), side = (i % 12) / 6 (0 left, 1 right), row r = i % 6 */
    const int g = lane / 12, k12 = lane - g * 12, side = k12 >= 6 ? 1 : 0, r = k12 - 6 * side;
    const int yy = min(max(yt + 4 * min(g, 3) - 1 + r, 0), height - 1);
    const int xx = side ? min(xt + 64, width - 1) : max(xt - 1, 0);
    rim_col = in[(size_t)yy * is + xx];
  }
  const bool any_edge = __any(type_raw == 2);
  const bool enabled = (nbmask_raw & 0x8000u) != 0;
  const int type = enabled ? type_raw : 0;
  const bool edge = type == 2;
#pragma unroll
  for (int r = 1; r <= 4; r++)
    if (r - 1 >= rows) {
#pragma unroll
      for (int k = 0; k < NW; k++) rw[r][k] = 0;
    }
#pragma unroll
  for (int k = 0; k < NW; k++) { rw[0][k] = 0; rw[5][k] = 0; }
  if (any_edge) {
    /* rows y0-1 and y0+4: the lanes 16 up / down hold them as their last / first own row; the wave tile's rim came from memory */
#pragma unroll
    for (int k = 0; k < NW; k++) {
      const uint32_t up = __shfl_up(rw[4][k], 16, 64), dn = __shfl_down(rw[1][k], 16, 64);
      rw[0][k] = ly > 0 ? up : ((valid && y0 > 0) ? rim_up[k] : 0u);
      rw[5][k] = ly < 3 ? dn : ((valid && y0 + 4 < height) ? rim_dn[k] : 0u);
    }
  }
  const int bd = c ? p.pp.bit_depth_chroma : p.pp.bit_depth_luma, maxv = (1 << bd) - 1;
  constexpr bool packed = PACKED;    /* bit depth <= 15 (the launcher decides): 16-bit lane differences need |c - a| < 32768 */
  /* rows as packed 16-bit pairs (s0,s1)(s2,s3) + the sample left / right of the block */
  uint32_t P[6][2];
  int Ls[6], Rs[6];
  int nb[6][6];
#pragma unroll
  for (int r = 0; r < 6; r++) {
    if (sizeof(PIX) == 2) { P[r][0] = rw[r][0]; P[r][1] = rw[r][NW - 1]; }
    else { P[r][0] = d_perm(0u, rw[r][0], 0x0c010c00u); P[r][1] = d_perm(0u, rw[r][0], 0x0c030c02u); }
    Ls[r] = Rs[r] = 0;
  }
  if (any_edge) {
#pragma unroll
    for (int r = 0; r < 6; r++) {
      const int yy = y0 - 1 + r;
      const bool yok = valid && yy >= 0 && yy < height;
      /* the wave tile's left / right rim sample of this lane's row group: fetched by lane ly * 12 + r (left) / + 6 + r (right);
         the tile lies at x = xt .. xt + 63, so x0 > 0 for an outer-left lane means xt > 0 */
      const int rimv = __shfl((int)rim_col, ly * 12 + (lx == 15 ? 6 : 0) + r, 64);
      int l = __shfl_up((int)(P[r][1] >> 16), 1, 64), rg = __shfl_down((int)(P[r][0] & 0xFFFFu), 1, 64);
      if (lx == 0) l = (edge && yok && x0 > 0) ? rimv : 0;
      if (lx == 15) rg = (edge && yok && x0 + 4 < width) ? rimv : 0;
      Ls[r] = l; Rs[r] = rg;
    }
  }
  if (!packed) {
#pragma unroll
    for (int r = 0; r < 6; r++) {
#pragma unroll
      for (int k = 0; k < 4; k++) nb[r][1 + k] = d_sao_sample<PIX>(rw[r], k);
      nb[r][0] = Ls[r]; nb[r][5] = Rs[r];
    }
  }
  /* pcm (with pcm_loop_filter_disable) / transquant-bypass samples are left alone (sao.cc:103-120); and is any
     neighbouring CTB unusable?  Both are rare: the wave takes the masked path only if some lane needs it */
  uint32_t skipmask = 0;
  if (valid && owned && type != 0 && (ctb.flags & M355_CTBF_HAS_PCM_OR_BYPASS)) {
    const bool plf = (p.pp.flags & M355_PF_PCM_LOOP_FILTER_DISABLE) != 0;
    for (int j = 0; j < 4; j++)
      for (int s2 = 0; s2 < 4; s2++) {
        const uint32_t ci = d_cu_index_at(p, min((x0 + s2) << csw, p.pp.width - 1), min((y0 + j) << csh, p.pp.height - 1));
        if (ci) {
          const m355_cu cu = p.cus[ci - 1];
          if ((plf && (cu.flags & M355_CUF_PCM)) || (cu.flags & M355_CUF_TRANSQUANT_BYPASS)) skipmask |= 1u << (j * 4 + s2);
        }
      }
  }
  const uint32_t nbmask = (valid && edge) ? (nbmask_raw & 0x1FFu) : 0u;
  const bool slow = __any(skipmask != 0 || nbmask != 0);
  if (!valid || !owned) return;

  if (packed) {
    uint32_t O[4][2];
#pragma unroll
    for (int j = 0; j < 4; j++) { O[j][0] = P[1 + j][0]; O[j][1] = P[1 + j][1]; }
    if (type != 0) {
      /* 5-entry offset table split into its positive and negative parts (bytes; |offset| <= 124): the sample becomes
         min(satsub(satadd(c, pos), neg), maxv) = Clip3(0, maxv, c + offset) on unsigned 16-bit lanes.
         edge: index = edgeIdx + 2 -> [o0, o1, 0, o2, o3] (sao.cc:95-100); band: index = min(k, 4) -> [o0..o3, 0] */
      const int t0 = so0, t1 = so1, t2 = edge ? 0 : so2, t3 = edge ? so2 : so3, t4 = edge ? so3 : 0;
      const uint32_t tpos_lo = (uint32_t)max(t0, 0) | ((uint32_t)max(t1, 0) << 8) | ((uint32_t)max(t2, 0) << 16) | ((uint32_t)max(t3, 0) << 24), tpos_hi = (uint32_t)max(t4, 0);
      const uint32_t tneg_lo = (uint32_t)max(-t0, 0) | ((uint32_t)max(-t1, 0) << 8) | ((uint32_t)max(-t2, 0) << 16) | ((uint32_t)max(-t3, 0) << 24), tneg_hi = (uint32_t)max(-t4, 0);
      uint32_t IDX[4][2];
      uint32_t bad = skipmask;
      if (edge) {
        const int cls = (ctb.sao_eo_class >> (2 * c)) & 3;
        const int xC = xCtb << l2w, yC = yCtb << l2h;
        const int nSW = 1 << l2w, nSH = 1 << l2h;
        const int ctbW_ = (xC + nSW > width) ? width - xC : nSW, ctbH_ = (yC + nSH > height) ? height - yC : nSH;
        const bool L = x0 == xC, R = x0 + 4 == xC + ctbW_, T = y0 == yC, Bt = y0 + 4 == yC + ctbH_;
        /* per row: (L,s0) / (s1,s2) / (s3,R) = the row seen one sample to the left / right */
        uint32_t Lp[6], Mp[6], Rp[6];
#pragma unroll
        for (int r = 0; r < 6; r++) {
          Lp[r] = (P[r][0] << 16) | (uint32_t)Ls[r];
          Mp[r] = __builtin_amdgcn_alignbit(P[r][1], P[r][0], 16);
          Rp[r] = __builtin_amdgcn_alignbit((uint32_t)Rs[r], P[r][1], 16);
        }
        /* class -> first neighbour offset (sao.cc:83-88): 0:(-1,0) 1:(0,-1) 2:(-1,-1) 3:(+1,-1); wave-uniform for luma */
        if (cls == 0) {
#pragma unroll
          for (int j = 0; j < 4; j++) { IDX[j][0] = d_sao_edge_idx(O[j][0], Lp[1 + j], Mp[1 + j]); IDX[j][1] = d_sao_edge_idx(O[j][1], Mp[1 + j], Rp[1 + j]); }
          if (slow) bad |= d_sao_bad<-1, 0>(L, R, T, Bt, nbmask);
        } else if (cls == 1) {
#pragma unroll
          for (int j = 0; j < 4; j++) { IDX[j][0] = d_sao_edge_idx(O[j][0], P[j][0], P[2 + j][0]); IDX[j][1] = d_sao_edge_idx(O[j][1], P[j][1], P[2 + j][1]); }
          if (slow) bad |= d_sao_bad<0, -1>(L, R, T, Bt, nbmask);
        } else if (cls == 2) {
#pragma unroll
          for (int j = 0; j < 4; j++) { IDX[j][0] = d_sao_edge_idx(O[j][0], Lp[j], Mp[2 + j]); IDX[j][1] = d_sao_edge_idx(O[j][1], Mp[j], Rp[2 + j]); }
          if (slow) bad |= d_sao_bad<-1, -1>(L, R, T, Bt, nbmask);
        } else {
#pragma unroll
          for (int j = 0; j < 4; j++) { IDX[j][0] = d_sao_edge_idx(O[j][0], Mp[j], Lp[2 + j]); IDX[j][1] = d_sao_edge_idx(O[j][1], Rp[j], Mp[2 + j]); }
          if (slow) bad |= d_sao_bad<1, -1>(L, R, T, Bt, nbmask);
        }
      } else {
        /* band offset (sao.cc:166-200): k = ((sample >> (bd-5)) - band_position) & 31 */
        const uint32_t bp2 = (uint32_t)band_pos * 0x00010001u;
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
          for (int q = 0; q < 2; q++) IDX[j][q] = d_pk_min_u16(d_pk_sub16(d_pk_lshr16(O[j][q], bd - 5), bp2) & 0x001F001Fu, 0x00040004u);
      }
      const uint32_t maxv2 = (uint32_t)maxv * 0x00010001u;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const uint32_t sel = d_perm(IDX[j][1], IDX[j][0], 0x06040200u);             /* the row's four table indices as bytes */
        const uint32_t pos4 = d_perm(tpos_hi, tpos_lo, sel), neg4 = d_perm(tneg_hi, tneg_lo, sel);
#pragma unroll
        for (int q = 0; q < 2; q++) {
          const uint32_t usel = q ? 0x0c030c02u : 0x0c010c00u;                       /* two bytes -> two 16-bit lanes */
          uint32_t vnew = d_pk_min_u16(d_pk_subsat_u16(d_pk_addsat_u16(O[j][q], d_perm(0u, pos4, usel)), d_perm(0u, neg4, usel)), maxv2);
          if (slow) {
            const uint32_t two = (bad >> (4 * j + 2 * q)) & 3u;
            const uint32_t keep = (((two & 1u) - 1u) & 0xFFFFu) | (((two >> 1) - 1u) << 16);   /* lanes that may be modified */
            vnew = (vnew & keep) | (O[j][q] & ~keep);
          }
          O[j][q] = vnew;
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
      PIX* q = out + (size_t)(y0 + j) * os + x0;
      if (sizeof(PIX) == 2) d_st_nt8(q, O[j][0], O[j][1]);
      else d_st_nt4(q, d_perm(O[j][1], O[j][0], 0x06040200u));
    }
    return;
  }

  PIX res[4][4];
#pragma unroll
  for (int j = 0; j < 4; j++)
#pragma unroll
    for (int s2 = 0; s2 < 4; s2++) res[j][s2] = (PIX)nb[1 + j][1 + s2];

  /* ---- bit depth 16: one sample per lane-op (the original formulation) ---- */
  if (type != 0) {
    if (edge) {
      const int cls = (ctb.sao_eo_class >> (2 * c)) & 3;
      const int xC = xCtb << l2w, yC = yCtb << l2h;
      const int nSW = 1 << l2w, nSH = 1 << l2h;
      const int ctbW_ = (xC + nSW > width) ? width - xC : nSW, ctbH_ = (yC + nSH > height) ? height - yC : nSH;
      const bool L = x0 == xC, R = x0 + 4 == xC + ctbW_, T = y0 == yC, Bt = y0 + rows == yC + ctbH_;
      /* class -> first neighbour offset (sao.cc:83-88): 0:(-1,0) 1:(0,-1) 2:(-1,-1) 3:(+1,-1); wave-uniform for luma */
      if (cls == 0) d_sao_edge_block<-1, 0, PIX>(so0, so1, so2, so3, nb, skipmask, L, R, T, Bt, nbmask, maxv, rows, res);
      else if (cls == 1) d_sao_edge_block<0, -1, PIX>(so0, so1, so2, so3, nb, skipmask, L, R, T, Bt, nbmask, maxv, rows, res);
      else if (cls == 2) d_sao_edge_block<-1, -1, PIX>(so0, so1, so2, so3, nb, skipmask, L, R, T, Bt, nbmask, maxv, rows, res);
      else d_sao_edge_block<1, -1, PIX>(so0, so1, so2, so3, nb, skipmask, L, R, T, Bt, nbmask, maxv, rows, res);
    } else {
      const unsigned btab = ((unsigned)so0 & 0xFFu) | (((unsigned)so1 & 0xFFu) << 8) | (((unsigned)so2 & 0xFFu) << 16) | (((unsigned)so3 & 0xFFu) << 24);
#pragma unroll
      for (int j = 0; j < 4; j++)
#pragma unroll
        for (int s2 = 0; s2 < 4; s2++) {
          const int cv = nb[1 + j][1 + s2];
          const int band = d_clip3(0, maxv, cv) >> (bd - 5);
          const int k = (band - band_pos) & 31;
          const int off = (int)(int8_t)d_byte_lookup(0u, btab, (unsigned)k & 3u);
          const int m = -(int)(k < 4 && !((skipmask >> (j * 4 + s2)) & 1u));
          res[j][s2] = (PIX)((d_clip3(0, maxv, cv + off) & m) | (cv & ~m));
        }
    }
  }
#pragma unroll
  for (int j = 0; j < 4; j++) {
    if (j >= rows) break;
    PIX* q = out + (size_t)(y0 + j) * os + x0;
    if (sizeof(PIX) == 2) d_st_nt8(q, (uint32_t)res[j][0] | ((uint32_t)res[j][1] << 16), (uint32_t)res[j][2] | ((uint32_t)res[j][3] << 16));
    else d_st_nt4(q, (uint32_t)res[j][0] | ((uint32_t)res[j][1] << 8) | ((uint32_t)res[j][2] << 16) | ((uint32_t)res[j][3] << 24));
  }
}

/* A 1-D grid whose blocks are dealt to the XCDs in runs of SAO_XCD_ROWS whole block ROWS (block b runs on XCD b % 8): the 256-sample blocks of a row — and the rows of a
   run — share one L2, so a row's rim lines (the 128-byte line left and right of every block, the rows above and below a run) are fetched from the fabric once instead of once
   per neighbour: a third less fabric fetch for this kernel, its time unchanged (profiles/r04_af_*, r06_v52_*).  Rows = luma rows, then the Cb rows, then the Cr rows (chroma
   rows are narrower: their surplus blocks leave at once). */
#define SAO_XCD_ROWS 2
template <class PIX, bool PACKED> __global__ void __launch_bounds__(256) k_sao(DevPic p, int gx, int gy0, int gy1)
{
  constexpr int RP = SAO_XCD_ROWS;
  const int b = (int)blockIdx.x, xcd = b & 7, i = b >> 3;
  const int x = i % gx, rr = i / gx;
  const int R = (rr / RP) * (8 * RP) + xcd * RP + rr % RP;
  if (R >= gy0 + 2 * gy1) return;
  const int c = R < gy0 ? 0 : (R < gy0 + gy1 ? 1 : 2), y = R - (c == 0 ? 0 : (c == 1 ? gy0 : gy0 + gy1));
  k_sao_body<PIX, PACKED>(p, c, x, y);
}
/* batch form: grid.z = 3 * picture + component */
template <class PIX, bool PACKED> __global__ void __launch_bounds__(256) k_sao_batch(DevBatch b)
{
  const int k = (int)blockIdx.z / 3, c = (int)blockIdx.z - 3 * k;
  M355_BATCH_PIC_AT(b, k);
  if (c && !p.pp.chroma_format_idc) return;
  k_sao_body<PIX, PACKED>(p, c, (int)blockIdx.x, (int)blockIdx.y);
}

void m355_launch_sao_batch(const HostBatch& b, bool hbd, hipStream_t st)
{
  int w = 0, h = 0; bool packed = true;
  for (int k = 0; k < b.n; k++) if ((b.on >> k) & 1u) {
    const DevPic& p = b.host[k];
    w = std::max(w, p.pw[0]); h = std::max(h, p.ph[0]);
    if (p.pp.bit_depth_luma > 15 || p.pp.bit_depth_chroma > 15) packed = false;
  }
  if (!w) return;
  const DevBatch d{b.dev, b.on};
  const dim3 grid((w + 255) / 256, (h + 15) / 16, 3 * b.n), block(256);
  if (!hbd) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_sao_batch<uint8_t, true>), grid, block, 0, st, d);
  else if (packed) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_sao_batch<uint16_t, true>), grid, block, 0, st, d);
  else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_sao_batch<uint16_t, false>), grid, block, 0, st, d);
}

void m355_launch_sao(const DevPic& p, bool hbd, hipStream_t st)
{
  /* one launch for all components (k_sao: block rows dealt to the XCDs); chroma blocks beyond the chroma plane exit at once */
  const int nc = p.pp.chroma_format_idc ? 3 : 1;
  const int gx = (p.pw[0] + 255) / 256, gy0 = (p.ph[0] + 15) / 16, gy1 = nc == 3 ? (p.ph[1] + 15) / 16 : 0;     /* 4 waves side by side: 256 x 16 samples */
  const int rows = gy0 + 2 * gy1, rows_pad = (rows + 8 * SAO_XCD_ROWS - 1) / (8 * SAO_XCD_ROWS) * (8 * SAO_XCD_ROWS);
  const dim3 grid((unsigned)(gx * rows_pad)), block(256);
  if (!hbd) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_sao<uint8_t, true>), grid, block, 0, st, p, gx, gy0, gy1);
  else if (p.pp.bit_depth_luma <= 15 && p.pp.bit_depth_chroma <= 15) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_sao<uint16_t, true>), grid, block, 0, st, p, gx, gy0, gy1);
  else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_sao<uint16_t, false>), grid, block, 0, st, p, gx, gy0, gy1);
}
