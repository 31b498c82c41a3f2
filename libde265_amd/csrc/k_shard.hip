/*
 * k_shard.hip — tile-boundary halo pack / unpack for a picture sharded by tiles across GPUs
 * (protocol in include/de265_mi355x.h, "Tile sharding").
 *
 * What crosses a tile boundary in the reference when loop_filter_across_tiles_enabled_flag is set:
 *   - deblocking of the boundary edge itself reads 4 luma (2 chroma) samples either side and the
 *     metadata of both units: edge flags / cbf (deblock.cc:132-227), PredMode, QP_Y, pcm / bypass,
 *     PBMotion (derive_boundaryStrength, deblock.cc:243-383);
 *   - the horizontal pass reads the vertical pass's output above / below a horizontal boundary
 *     (deblock.cc:919-939 runs all vertical edges first);
 *   - SAO edge classes read the deblocked 1-sample ring incl. the diagonal corner (sao.cc:83-88).
 * Every buffer has a canonical, rank-independent layout (HaloLayout) in which each element is owned
 * by exactly one rank (the owner of the CTB it lies in); pack writes the owned elements and ZERO for
 * all others, so an integer SUM all-reduce over 32-bit words — or a plain owner -> neighbour copy —
 * completes it.  unpack writes only elements of NON-owned CTBs, i.e. it never touches own samples.
 * Foreign border units are injected as records appended to the local cus[] / pbs[] lists, so
 * k_deblock reads them through the same planes as local units.
 * Roofline: trivial traffic (a few hundred kB per picture); these kernels are latency-bound.
 */
#include "k_common.h"

struct SamplePos { int c, x, y; };

/* canonical element index -> (component, x, y); `which` bit 0: column strips present, bit 1: row strips */
__device__ __forceinline__ bool d_halo_locate(const DevPic& p, const HaloLayout& h, int which, int i, SamplePos* o)
{
  const int ncol = (which & 1) ? h.col_ofs[3] : 0;
  if (i < ncol) {
    const int c = i >= h.col_ofs[2] ? 2 : (i >= h.col_ofs[1] ? 1 : 0);
    const int l = i - h.col_ofs[c], w2 = 2 * h.hw[c];
    const int k = l % w2, t = l / w2;
    const int y = t % p.ph[c], b = t / p.ph[c];
    o->c = c; o->x = h.xb[c][b] - h.hw[c] + k; o->y = y;
    return true;
  }
  i -= ncol;
  if (!(which & 2) || i >= h.row_ofs[3]) return false;
  const int c = i >= h.row_ofs[2] ? 2 : (i >= h.row_ofs[1] ? 1 : 0);
  const int l = i - h.row_ofs[c];
  const int x = l % p.pw[c], t = l / p.pw[c];
  const int h2 = 2 * h.hh[c];
  const int k = t % h2, b = t / h2;
  o->c = c; o->x = x; o->y = h.yb[c][b] - h.hh[c] + k;
  return true;
}

__device__ __forceinline__ bool d_owned_sample(const DevPic& p, int c, int x, int y)
{
  const int xl = c ? x * p.sw : x, yl = c ? y * p.sh : y;
  return p.ctb_owner[d_ctb_of(p, min(xl, p.pp.width - 1), min(yl, p.pp.height - 1))] != 0;
}

template <class PIX>
__global__ void __launch_bounds__(256) k_halo_pack_samples(DevPic p, HaloLayout h, int which, int n, int n_padded, PIX* buf)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_padded) return;
  PIX v = 0;
  SamplePos s;
  if (i < n && d_halo_locate(p, h, which, i, &s) && d_owned_sample(p, s.c, s.x, s.y))
    v = ((const PIX*)p.plane[s.c])[(size_t)s.y * p.stride[s.c] + s.x];
  buf[i] = v;
}

template <class PIX>
__global__ void __launch_bounds__(256) k_halo_unpack_samples(DevPic p, HaloLayout h, int which, int n, const PIX* buf)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  SamplePos s;
  if (d_halo_locate(p, h, which, i, &s) && !d_owned_sample(p, s.c, s.x, s.y))
    ((PIX*)p.plane[s.c])[(size_t)s.y * p.stride[s.c] + s.x] = buf[i];
}

/* border unit j -> 4x4 unit coordinates: vertical boundaries first ((b, side, y4)), then horizontal ((b, side, x4)) */
__device__ __forceinline__ void d_halo_unit(const DevPic& p, const HaloLayout& h, int j, int* x4, int* y4)
{
  const int nv = 2 * h.n_vb * p.h4;
  if (j < nv) {
    const int y = j % p.h4, t = j / p.h4;
    *x4 = (h.xb[0][t >> 1] >> 2) - 1 + (t & 1); *y4 = y;
  } else {
    j -= nv;
    const int x = j % p.w4, t = j / p.w4;
    *x4 = x; *y4 = (h.yb[0][t >> 1] >> 2) - 1 + (t & 1);
  }
}

/* record (4 words): w0 = edge bits | pred_mode << 8 | cu flags << 16 | pb flags << 24 (bit 31: has PB)
 *                   w1 = qp_y (u8) | ref_slot[0] << 8 | ref_slot[1] << 16 | valid << 24;  w2, w3 = mv[0], mv[1] */
__global__ void __launch_bounds__(256) k_halo_pack_meta(DevPic p, HaloLayout h, uint4* rec)
{
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= h.n_units) return;
  int x4, y4;
  d_halo_unit(p, h, j, &x4, &y4);
  uint4 r = make_uint4(0, 0, 0, 0);
  const int xl = x4 << 2, yl = y4 << 2;
  if (p.ctb_owner[d_ctb_of(p, xl, yl)]) {
    const uint32_t ci = d_cu_index_at(p, xl, yl);
    if (ci) {
      const m355_cu cu = p.cus[ci - 1];
      const int u = y4 * p.w4 + x4;
      const uint32_t edge = (uint32_t)(p.edge_tu[u] | p.edge_pb[u]);
      r.x = edge | ((uint32_t)cu.pred_mode << 8) | ((uint32_t)cu.flags << 16);
      r.y = ((uint32_t)(uint8_t)cu.qp_y) | (1u << 24);
      if (cu.pred_mode != 0) {
        const uint32_t pi = p.pb_of[u];
        if (pi) {
          const m355_pb pb = p.pbs[pi - 1];
          r.x |= ((uint32_t)(pb.flags & 0x7F) | 0x80u) << 24;
          r.y |= ((uint32_t)(uint8_t)pb.ref_slot[0] << 8) | ((uint32_t)(uint8_t)pb.ref_slot[1] << 16);
          r.z = (uint32_t)(uint16_t)pb.mv[0][0] | ((uint32_t)(uint16_t)pb.mv[0][1] << 16);
          r.w = (uint32_t)(uint16_t)pb.mv[1][0] | ((uint32_t)(uint16_t)pb.mv[1][1] << 16);
        }
      }
    }
  }
  rec[j] = r;
}

__global__ void __launch_bounds__(256) k_halo_unpack_meta(DevPic p, HaloLayout h, const uint4* rec)
{
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= h.n_units) return;
  int x4, y4;
  d_halo_unit(p, h, j, &x4, &y4);
  const int xl = x4 << 2, yl = y4 << 2;
  if (p.ctb_owner[d_ctb_of(p, xl, yl)]) return;          /* own unit: local metadata is authoritative */
  const uint4 r = rec[j];
  if (!(r.y >> 24)) return;                               /* nobody covers the unit */
  const int u = y4 * p.w4 + x4;
  const int l2m = p.pp.log2_min_cb_size;
  m355_cu cu;
  cu.x = (uint16_t)((xl >> l2m) << l2m); cu.y = (uint16_t)((yl >> l2m) << l2m);
  cu.log2_size = (uint8_t)l2m; cu.pred_mode = (uint8_t)((r.x >> 8) & 0xFF); cu.part_mode = 0;
  cu.qp_y = (int8_t)(r.y & 0xFF); cu.flags = (uint8_t)((r.x >> 16) & 0xFF);
  cu.reserved[0] = cu.reserved[1] = cu.reserved[2] = 0;
  /* several units of one min CB write records with identical contents: whichever index lands is fine */
  ((m355_cu*)p.cus)[p.halo_cu_base + j] = cu;
  p.cb_cu[(yl >> l2m) * p.wcb + (xl >> l2m)] = (uint32_t)(p.halo_cu_base + j) + 1;
  p.edge_tu[u] = (uint8_t)(r.x & (E_TU_V | E_TU_H | E_NONZERO));
  p.edge_pb[u] = (uint8_t)(r.x & (E_PB_V | E_PB_H));
  uint32_t pi = 0;
  if (r.x >> 31) {
    m355_pb pb;
    pb.x = (uint16_t)xl; pb.y = (uint16_t)yl; pb.w = pb.h = 4; pb.flags = (uint8_t)((r.x >> 24) & 0x7F); pb.reserved = 0;
    pb.ref_slot[0] = (int8_t)((r.y >> 8) & 0xFF); pb.ref_slot[1] = (int8_t)((r.y >> 16) & 0xFF);
    pb.mv[0][0] = (int16_t)(r.z & 0xFFFF); pb.mv[0][1] = (int16_t)(r.z >> 16);
    pb.mv[1][0] = (int16_t)(r.w & 0xFFFF); pb.mv[1][1] = (int16_t)(r.w >> 16);
    pb.wt_idx[0] = pb.wt_idx[1] = 0; pb.reserved2 = 0;
    ((m355_pb*)p.pbs)[p.halo_pb_base + j] = pb;
    pi = (uint32_t)(p.halo_pb_base + j) + 1;
  }
  p.pb_of[u] = pi;
}

static int halo_samples(const HaloLayout& h, int which) { return ((which & 1) ? h.col_ofs[3] : 0) + ((which & 2) ? h.row_ofs[3] : 0); }

void m355_launch_halo_pack(const DevPic& p, const HaloLayout& h, bool hbd, int which, void* samples, uint32_t* meta, hipStream_t st)
{
  if (meta && h.n_units) hipLaunchKernelGGL(k_halo_pack_meta, dim3((h.n_units + 255) / 256), dim3(256), 0, st, p, h, (uint4*)meta);
  const int n = halo_samples(h, which);
  if (!n) return;
  const int per_word = hbd ? 2 : 4, n_padded = (n + per_word - 1) / per_word * per_word;   /* whole 32-bit words */
  if (hbd) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_halo_pack_samples<uint16_t>), dim3((n_padded + 255) / 256), dim3(256), 0, st, p, h, which, n, n_padded, (uint16_t*)samples);
  else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_halo_pack_samples<uint8_t>), dim3((n_padded + 255) / 256), dim3(256), 0, st, p, h, which, n, n_padded, (uint8_t*)samples);
}

void m355_launch_halo_unpack(const DevPic& p, const HaloLayout& h, bool hbd, int which, const void* samples, const uint32_t* meta, hipStream_t st)
{
  if (meta && h.n_units) hipLaunchKernelGGL(k_halo_unpack_meta, dim3((h.n_units + 255) / 256), dim3(256), 0, st, p, h, (const uint4*)meta);
  const int n = halo_samples(h, which);
  if (!n) return;
  if (hbd) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_halo_unpack_samples<uint16_t>), dim3((n + 255) / 256), dim3(256), 0, st, p, h, which, n, (const uint16_t*)samples);
  else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_halo_unpack_samples<uint8_t>), dim3((n + 255) / 256), dim3(256), 0, st, p, h, which, n, (const uint8_t*)samples);
}

/* ---- finished tiles <-> the all-gather buffer X3: every rectangle of a call in ONE launch (a picture with many tiles would
 * otherwise cost a 2-D copy per tile and plane; the launches, not the bytes, were what phases 3 / 4 took) ---- */
template <int V>
__global__ void __launch_bounds__(256) k_tiles_copy(TileCopyArgs a, char* xbuf, int to_slot)
{
  const TileCopyRect r = a.r[blockIdx.y];
  const unsigned upr = r.wb / V;                               /* V-byte units per row */
  const unsigned long long total = (unsigned long long)upr * r.h;
  char* plane = a.plane[r.plane] + (size_t)r.y * a.pitch[r.plane] + r.xb;
  char* slot = xbuf + r.ofs;
  for (unsigned long long u = (unsigned long long)blockIdx.x * 256 + threadIdx.x; u < total; u += (unsigned long long)gridDim.x * 256) {
    const unsigned row = (unsigned)(u / upr), col = (unsigned)(u - (unsigned long long)row * upr);
    char* pf = plane + (size_t)row * a.pitch[r.plane] + (size_t)col * V;
    char* ps = slot + (size_t)row * r.wb + (size_t)col * V;
    if (V == 16) { if (to_slot) *(uint4*)ps = *(const uint4*)pf; else *(uint4*)pf = *(const uint4*)ps; }
    else { if (to_slot) *(uint32_t*)ps = *(const uint32_t*)pf; else *(uint32_t*)pf = *(const uint32_t*)ps; }
  }
}

void m355_launch_tiles_copy(const TileCopyArgs& a, int n, void* xbuf, bool to_slot, hipStream_t st)
{
  if (n <= 0) return;
  bool wide = ((uintptr_t)xbuf & 15) == 0;
  unsigned long long most = 0;
  for (int i = 0; i < n; i++) {
    const TileCopyRect& r = a.r[i];
    if ((r.wb | r.xb | r.ofs) & 15) wide = false;
    const unsigned long long b = (unsigned long long)r.wb * r.h;
    if (b > most) most = b;
  }
  for (int c = 0; c < 3; c++) if (((uintptr_t)a.plane[c] | a.pitch[c]) & 15) wide = false;
  const int V = wide ? 16 : 4;
  unsigned gx = (unsigned)((most / V + 256 * 8 - 1) / (256 * 8));      /* ~8 units per thread */
  if (gx < 1) gx = 1;
  if (wide) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_tiles_copy<16>), dim3(gx, n), dim3(256), 0, st, a, (char*)xbuf, to_slot ? 1 : 0);
  else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_tiles_copy<4>), dim3(gx, n), dim3(256), 0, st, a, (char*)xbuf, to_slot ? 1 : 0);
}

/* halo exchange over point-to-point copies: every element has exactly one producer (zero elsewhere), so adding the peers' buffers
   completes this rank's copy */
__global__ void __launch_bounds__(256) k_halo_add(uint32_t* buf, const uint32_t* scratch, uint32_t pitch_words, int n, uint32_t words)
{
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= words) return;
  uint32_t v = buf[i];
  for (int k = 0; k < n; k++) v += scratch[(size_t)k * pitch_words + i];
  buf[i] = v;
}
void m355_launch_halo_add(uint32_t* buf, const uint32_t* scratch, uint32_t pitch_words, int n, uint32_t words, hipStream_t st)
{
  if (!words || !n) return;
  hipLaunchKernelGGL(k_halo_add, dim3((words + 255) / 256), dim3(256), 0, st, buf, scratch, pitch_words, n, words);
}
