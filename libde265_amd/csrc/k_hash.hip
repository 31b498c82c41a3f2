/*
 * k_hash.hip — SEI decoded-picture-hash of a device frame without copying it back.
 *
 * Replaces compute_CRC_8bit_fast (sei.cc:209-233) and compute_checksum (sei.cc:161-186) as used by
 * process_sei_decoded_picture_hash (sei.cc:276-356); the message is the plane row by row, samples of > 8 bit as two
 * bytes, low byte first (raw_hash_data::prepare_16bit, sei.cc:141-158) — the device planes' own byte order.
 *
 * CRC: the reference's byte update (crc_process_byte_parallel, sei.cc:198-207) is the direct form of CRC-16/CCITT
 * (P = x^16+x^12+x^5+1), which is linear over GF(2):   state = init * x^(8N) + m(x) * x^16   (mod P)
 * for an N-byte message m.  So the plane is cut into spans of rows, one wavefront per span: each lane runs the byte
 * update over 16 adjacent bytes from state 0, its result is multiplied by x^(8 * bytes that follow it in the 1 KB block)
 * (a compile-time table for full blocks), the wave XOR-reduces, blocks are chained Horner-style
 * (acc = acc * x^(8 * block_len) + block), and the span's result is multiplied by x^(8 * bytes after the span) and
 * XOR-ed into the plane's accumulator.  The host adds init * x^(8N), init = the reference's 0xFFFF run through two zero
 * bytes.  Reads are 16-byte coalesced; HBM-bound on paper (every byte read once), in practice ~350 VALU instructions per
 * KB per wave.
 *
 * Checksum: sum over message bytes of (byte ^ xorMask(x, y)), xorMask = (x & 0xFF) ^ (y & 0xFF) ^ (x >> 8) ^ (y >> 8),
 * modulo 2^32 — a plain reduction.  For > 8-bit planes this is the H.265 D.3.19 definition; the reference's
 * compute_checksum halves an already sample-unit stride there (sei.cc:174), so on such planes it hashes the wrong rows
 * and is not a usable oracle — DESIGN.md §5.
 *
 * MD5 is one serial chain per plane (3 per picture) — nothing for 256 CUs to do; m355_frame_hash downloads the
 * planes and hashes them on host threads (runtime.hip).
 */
#include <string.h>

#include "k_common.h"
#include "k_hash.h"

/* ---- GF(2)[x] / P, 16-bit residues ---- */
__host__ __device__ constexpr uint32_t gf_mul(uint32_t a, uint32_t b)
{
  uint32_t r = 0;
  for (int i = 15; i >= 0; i--) {
    r <<= 1;
    if (r & 0x10000u) r ^= 0x11021u;
    if ((b >> i) & 1u) r ^= a;
  }
  return r;
}
__host__ __device__ constexpr uint32_t gf_pow_x8(uint64_t nbytes)   /* x^(8 * nbytes) */
{
  uint32_t base = 0x100u, res = 1u;
  while (nbytes) {
    if (nbytes & 1u) res = gf_mul(res, base);
    base = gf_mul(base, base);
    nbytes >>= 1;
  }
  return res;
}
uint32_t m355_crc_init_term(uint64_t nbytes)
{
  /* 0xFFFF through two zero bytes (sei.cc:215-218) = 0xFFFF * x^16 */
  return gf_mul(gf_mul(0xFFFFu, gf_pow_x8(2)), gf_pow_x8(nbytes));
}

#define HASH_BLOCK 1024   /* bytes per wave step: 64 lanes x 16 */
struct HashTables { uint16_t after[64]; uint16_t blk; uint16_t sq[40]; };   /* sq[k] = x^(8 * 2^k) */
constexpr HashTables make_hash_tables()
{
  HashTables t{};
  for (int l = 0; l < 64; l++) t.after[l] = (uint16_t)gf_pow_x8((uint64_t)(63 - l) * 16);
  t.blk = (uint16_t)gf_pow_x8(HASH_BLOCK);
  for (int k = 0; k < 40; k++) t.sq[k] = (uint16_t)gf_pow_x8((uint64_t)1 << k);
  return t;
}
__constant__ HashTables c_hash = make_hash_tables();

/* x^(8 * nbytes) on the device: one multiplication per set bit of nbytes (nbytes < 2^40) */
__device__ __forceinline__ uint32_t d_pow_x8(uint64_t nbytes)
{
  uint32_t res = 1u;
  for (int k = 0; nbytes; k++, nbytes >>= 1)
    if (nbytes & 1u) res = gf_mul(res, c_hash.sq[k]);
  return res;
}
__device__ __forceinline__ uint32_t d_crc_byte(uint32_t crc, uint32_t byte)
{
  const uint32_t s = byte ^ (crc >> 8);
  const uint32_t t = s ^ (s >> 4);
  return ((crc << 8) ^ t ^ (t << 5) ^ (t << 12)) & 0xFFFFu;
}
__device__ __forceinline__ uint32_t d_wave_xor(uint32_t v)
{
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v ^= __shfl_xor(v, m, 64);
  return v;
}
__device__ __forceinline__ uint32_t d_wave_add(uint32_t v)
{
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}

/* the lane's (up to) 16 bytes of a row at byte offset o; rows start 128-byte aligned, o is a multiple of 16 */
__device__ __forceinline__ void d_load16(const uint8_t* row, int o, int n, uint32_t w[4])
{
  w[0] = w[1] = w[2] = w[3] = 0;
  if (n == 16) { const uint4 v = *(const uint4*)(row + o); w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w; }
  else
    for (int k = 0; k < n; k++) w[k >> 2] |= (uint32_t)row[o + k] << (8 * (k & 3));
}

/* one wave per (plane, span of rows); wave index -> plane by the prefix counts in a.first[] */
template <int TYPE>
__global__ void __launch_bounds__(256) k_frame_hash(HashArgs a)
{
  const int lane = threadIdx.x & 63;
  const int wv = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wv >= a.first[3]) return;
  const int c = wv >= a.first[2] ? 2 : (wv >= a.first[1] ? 1 : 0);
  const HashPlane pl = a.pl[c];
  const int y0 = (wv - a.first[c]) * a.rows_per_wave, y1 = min(y0 + a.rows_per_wave, pl.h);

  if (TYPE == M355_HASH_CRC) {
    uint32_t acc = 0;
    for (int y = y0; y < y1; y++) {
      const uint8_t* row = pl.base + (size_t)y * pl.pitch;
      for (int o = 0; o < pl.row_bytes; o += HASH_BLOCK) {
        const int len = min(HASH_BLOCK, pl.row_bytes - o);
        const int lo = lane * 16, n = d_clip3(0, 16, len - lo);
        uint32_t w[4];
        d_load16(row, o + lo, n, w);
        uint32_t crc = 0;
#pragma unroll
        for (int k = 0; k < 16; k++)
          if (k < n) crc = d_crc_byte(crc, (w[k >> 2] >> (8 * (k & 3))) & 0xFFu);
        uint32_t mult, step;
        if (len == HASH_BLOCK) { mult = c_hash.after[lane]; step = c_hash.blk; }
        else { mult = d_pow_x8((uint64_t)max(len - lo - n, 0)); step = d_pow_x8((uint64_t)len); }   /* last block of a row */
        const uint32_t blk = d_wave_xor(n > 0 ? gf_mul(crc, mult) : 0u);
        acc = gf_mul(acc, step) ^ blk;
      }
    }
    const uint64_t after = (uint64_t)(pl.h - y1) * (uint64_t)pl.row_bytes;
    const uint32_t v = gf_mul(acc, d_pow_x8(after));
    if (lane == 0 && y1 > y0) atomicXor(&a.out[c], v);
  } else {
    uint32_t sum = 0;
    for (int y = y0; y < y1; y++) {
      const uint8_t* row = pl.base + (size_t)y * pl.pitch;
      const uint32_t my = (uint32_t)((y & 0xFF) ^ (y >> 8));
      for (int o = lane * 16; o < pl.row_bytes; o += HASH_BLOCK) {
        const int n = min(16, pl.row_bytes - o);
        uint32_t w[4];
        d_load16(row, o, n, w);
#pragma unroll
        for (int k = 0; k < 16; k++)
          if (k < n) {
            const int x = (o + k) >> (pl.bpp - 1);            /* sample column: both bytes of a 16-bit sample share the mask */
            const uint32_t m = (my ^ (uint32_t)(x & 0xFF) ^ (uint32_t)(x >> 8)) & 0xFFu;
            sum += ((w[k >> 2] >> (8 * (k & 3))) & 0xFFu) ^ m;
          }
      }
    }
    sum = d_wave_add(sum);
    if (lane == 0 && y1 > y0) atomicAdd(&a.out[c], sum);
  }
}

void m355_launch_frame_hash(const HashArgs& a, int type, hipStream_t st)
{
  const int nw = a.first[3];
  if (!nw) return;
  if (type == M355_HASH_CRC) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_frame_hash<M355_HASH_CRC>), dim3((nw + 3) / 4), dim3(256), 0, st, a);
  else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_frame_hash<M355_HASH_CHECKSUM>), dim3((nw + 3) / 4), dim3(256), 0, st, a);
}

/* ---- MD5 (RFC 1321) on the host: a serial chain, see the header ---- */
namespace {
struct Md5 {
  uint32_t s[4] = {0x67452301u, 0xefcdab89u, 0x98badcfeu, 0x10325476u};
  uint64_t n = 0;
  uint8_t buf[64];
  static uint32_t rol(uint32_t v, int r) { return (v << r) | (v >> (32 - r)); }
  void block(const uint8_t* p)
  {
    static const uint32_t K[64] = {
        0xd76aa478, 0xe8c7b756, 0x242070db, 0xc1bdceee, 0xf57c0faf, 0x4787c62a, 0xa8304613, 0xfd469501, 0x698098d8, 0x8b44f7af, 0xffff5bb1,
        0x895cd7be, 0x6b901122, 0xfd987193, 0xa679438e, 0x49b40821, 0xf61e2562, 0xc040b340, 0x265e5a51, 0xe9b6c7aa, 0xd62f105d, 0x02441453,
        0xd8a1e681, 0xe7d3fbc8, 0x21e1cde6, 0xc33707d6, 0xf4d50d87, 0x455a14ed, 0xa9e3e905, 0xfcefa3f8, 0x676f02d9, 0x8d2a4c8a, 0xfffa3942,
        0x8771f681, 0x6d9d6122, 0xfde5380c, 0xa4beea44, 0x4bdecfa9, 0xf6bb4b60, 0xbebfbc70, 0x289b7ec6, 0xeaa127fa, 0xd4ef3085, 0x04881d05,
        0xd9d4d039, 0xe6db99e5, 0x1fa27cf8, 0xc4ac5665, 0xf4292244, 0x432aff97, 0xab9423a7, 0xfc93a039, 0x655b59c3, 0x8f0ccc92, 0xffeff47d,
        0x85845dd1, 0x6fa87e4f, 0xfe2ce6e0, 0xa3014314, 0x4e0811a1, 0xf7537e82, 0xbd3af235, 0x2ad7d2bb, 0xeb86d391};
    static const int R[64] = {7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 5, 9,  14, 20, 5, 9,  14, 20, 5, 9,  14, 20, 5, 9,  14, 20,
                              4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21};
    uint32_t m[16];
    for (int i = 0; i < 16; i++) m[i] = (uint32_t)p[4 * i] | ((uint32_t)p[4 * i + 1] << 8) | ((uint32_t)p[4 * i + 2] << 16) | ((uint32_t)p[4 * i + 3] << 24);
    uint32_t a = s[0], b = s[1], c = s[2], d = s[3];
    for (int i = 0; i < 64; i++) {
      uint32_t f;
      int g;
      if (i < 16) { f = (b & c) | (~b & d); g = i; }
      else if (i < 32) { f = (d & b) | (~d & c); g = (5 * i + 1) & 15; }
      else if (i < 48) { f = b ^ c ^ d; g = (3 * i + 5) & 15; }
      else { f = c ^ (b | ~d); g = (7 * i) & 15; }
      const uint32_t t = d;
      d = c; c = b;
      b = b + rol(a + f + K[i] + m[g], R[i]);
      a = t;
    }
    s[0] += a; s[1] += b; s[2] += c; s[3] += d;
  }
  void update(const uint8_t* p, size_t len)
  {
    size_t fill = (size_t)(n & 63);
    n += len;
    if (fill) {
      const size_t take = len < 64 - fill ? len : 64 - fill;
      memcpy(buf + fill, p, take);
      p += take; len -= take; fill += take;
      if (fill < 64) return;
      block(buf);
    }
    for (; len >= 64; p += 64, len -= 64) block(p);
    if (len) memcpy(buf, p, len);
  }
  void final(uint8_t out[16])
  {
    const uint64_t bits = n * 8;
    const uint8_t pad = 0x80, zero = 0;
    update(&pad, 1);
    while ((n & 63) != 56) update(&zero, 1);
    uint8_t lenb[8];
    for (int i = 0; i < 8; i++) lenb[i] = (uint8_t)(bits >> (8 * i));
    update(lenb, 8);
    for (int i = 0; i < 4; i++)
      for (int k = 0; k < 4; k++) out[4 * i + k] = (uint8_t)(s[i] >> (8 * k));
  }
};
}   // namespace

void m355_md5_rows(const uint8_t* data, size_t pitch, int row_bytes, int h, uint8_t out[16])
{
  Md5 md;
  for (int y = 0; y < h; y++) md.update(data + (size_t)y * pitch, (size_t)row_bytes);
  md.final(out);
}
