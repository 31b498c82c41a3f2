// The MFMA decision (BASELINE.json north_star: "MFMA used only if a batched-DCT-as-GEMM formulation actually wins on rocprof").
// 32x32 inverse DCT + add (transform_idct_add, fallback-dct.cc:550-691; matrix :512-545) of N dense int16 blocks onto a 10-bit plane:
//   k_dot2 : the product's scheme (k_residual.hip d_rb_compute): lane = (column, block), two blocks per wave, coefficient tile in LDS as
//            vertical int16 pairs, column pass and row pass as v_dot2c_i32_i16 with the matrix pair as a SCALAR operand, pruned to the
//            occupied rows / columns (`occ`), first-stage clip to int16, 16-byte row accesses to the plane.
//   k_mfma : one block per wave, two GEMMs on the matrix pipe with v_mfma_i32_32x32x32_i8, exact in int32:
//            G^T = C^T x M     with the int16 operand split into a signed high byte and a low byte biased by -128 (two MFMAs, the
//            Out = G x M        accumulators joined as (hi << 8) + lo; the bias is 128 x the matrix's column sum = ONE constant per lane),
//            the first-stage values stay in the lane that computed them: the k-slot order of the second GEMM's A operand is whatever the
//            first one's C/D layout left (row = (v & 3) + 8 (v >> 2) + 4 (lane >> 5)), and the constant B operand is permuted to match —
//            no LDS, no transpose between the passes.  Coefficients are read transposed (the product scatters sparse (pos, level) pairs
//            into its tile, so the tile's layout is free).
// Both are checked against a plain C restatement of the reference on the host, bit for bit.
//   hipcc -O3 --offload-arch=gfx950 -o ub_mfma_idct tools/ubench/ub_mfma_idct.hip && ./ub_mfma_idct
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int qw(int m)
{
  constexpr int t[33] = {64, 90, 90, 90, 89, 88, 87, 85, 83, 82, 80, 78, 75, 73, 70, 67, 64, 61, 57, 54, 50, 46, 43, 38, 36, 31, 25, 22, 18, 13, 9, 4, 0};
  m &= 127;
  return m <= 32 ? t[m] : (m <= 64 ? -t[64 - m] : (m < 96 ? -t[m - 64] : t[128 - m]));
}
constexpr int M32(int j, int i) { return qw(j * (2 * i + 1)); }   // mat_dct[j][i]

struct Tab { uint32_t pair[16 * 32]; };    // (M[2q][i], M[2q+1][i]) as int16 pairs, [q * 32 + i]
constexpr Tab make_tab()
{
  Tab t{};
  for (int q = 0; q < 16; q++)
    for (int i = 0; i < 32; i++) t.pair[q * 32 + i] = ((uint32_t)M32(2 * q, i) & 0xFFFFu) | ((uint32_t)M32(2 * q + 1, i) << 16);
  return t;
}
__constant__ Tab c_tab = make_tab();

#define PITCH 4096          // samples per plane row
#define BLK_X (PITCH / 32)
typedef short short2_ __attribute__((ext_vector_type(2)));
typedef int int4_ __attribute__((ext_vector_type(4)));
typedef int int16_ __attribute__((ext_vector_type(16)));
__device__ __forceinline__ int dot2(unsigned a, unsigned b, int c) { return __builtin_amdgcn_sdot2(__builtin_bit_cast(short2_, a), __builtin_bit_cast(short2_, b), c, false); }
__device__ __forceinline__ int clip3(int lo, int hi, int v) { return min(max(v, lo), hi); }

// ---- the product's scheme ----
__global__ void __launch_bounds__(64) k_dot2(const uint32_t* __restrict__ pairs, uint16_t* __restrict__ plane, int nblk, int occ, int bd)
{
  constexpr int NT = 32, QN = 16, GP = QN + 1;
  __shared__ uint32_t smem[2 * (QN * NT + NT * GP)];
  const int lane = threadIdx.x, c = lane & 31, b = lane >> 5;
  const int blk = blockIdx.x * 2 + b;
  uint32_t* cfp = smem + b * (QN * NT + NT * GP);
  uint32_t* gp = cfp + QN * NT;
  int16_t* g16 = (int16_t*)gp;
  const bool active = blk < nblk;
  const uint32_t* src = pairs + (size_t)(active ? blk : 0) * (QN * NT);
  const int q1 = (occ + 1) >> 1;
  for (int q = 0; q < q1; q++) cfp[q * NT + c] = src[q * NT + c];
  __builtin_amdgcn_wave_barrier();
  int acc[NT];
#pragma unroll
  for (int i = 0; i < NT; i++) acc[i] = 64;
  for (int q = 0; q < q1; q++) {
    const uint32_t v = cfp[q * NT + c];
#pragma unroll
    for (int i = 0; i < NT; i++) acc[i] = dot2(v, c_tab.pair[q * NT + i], acc[i]);
  }
#pragma unroll
  for (int i = 0; i < NT; i++) g16[(i * GP) * 2 + c] = (int16_t)clip3(-32768, 32767, acc[i] >> 7);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const int post = 20 - bd;
#pragma unroll
  for (int i = 0; i < NT; i++) acc[i] = 1 << (post - 1);
  for (int q = 0; q < q1; q++) {
    const uint32_t v = gp[c * GP + q];
#pragma unroll
    for (int i = 0; i < NT; i++) acc[i] = dot2(v, c_tab.pair[q * NT + i], acc[i]);
  }
  if (!active) return;
  uint16_t* d = plane + (size_t)((blk / BLK_X) * 32 + c) * PITCH + (blk % BLK_X) * 32;
  const int maxv = (1 << bd) - 1;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    uint4 w = *(const uint4*)(d + 8 * k);
    uint32_t* ww = (uint32_t*)&w;
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const int lo = clip3(0, maxv, (int)(ww[e] & 0xFFFF) + (acc[8 * k + 2 * e] >> post)), hi = clip3(0, maxv, (int)(ww[e] >> 16) + (acc[8 * k + 2 * e + 1] >> post));
      ww[e] = (uint32_t)lo | ((uint32_t)hi << 16);
    }
    *(uint4*)(d + 8 * k) = w;
  }
}

// ---- the matrix pipe ----
// constant operands, per lane (built once per wave): B1[s] = M[16 g + s][i], B2[s] = M[col(s, g)][i] with col(s, g) = (s & 3) + 8 (s >> 2) + 4 g, i = lane & 31, g = lane >> 5
__global__ void __launch_bounds__(256) k_mfma(const int16_t* __restrict__ ct, uint16_t* __restrict__ plane, int nblk, int bd, int per_wave, const int* __restrict__ lane_tab)
{
  const int lane = threadIdx.x & 63, i = lane & 31, g = lane >> 5;
  const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
  /* the lane's constant operands from a 64 x 12-word table (the host built it: a kernel of the product would keep it in constant memory) */
  const int4_ B1 = *(const int4_*)(lane_tab + lane * 12), B2 = *(const int4_*)(lane_tab + lane * 12 + 4);
  const int cs = lane_tab[lane * 12 + 8];
  const int post = 20 - bd;
  const unsigned maxv = ((1u << bd) - 1u) * 0x10001u;
  const unsigned sel = (lane & 1) ? 0x07060302u : 0x05040100u;    // odd lanes: (nbr.hi, own.hi); even lanes: (own.lo, nbr.lo)  [perm(hi = second, lo = first)]
  for (int it = 0; it < per_wave; it++) {
    const int blk = wave * per_wave + it;
    if (blk >= nblk) break;
    // A = C^T: row m = col = lane & 31, k slot s <-> j = 16 g + s : 32 contiguous bytes of the transposed tile
    const uint4* src = (const uint4*)(ct + (size_t)blk * 1024 + i * 32 + g * 16);
    const uint4 c0 = src[0], c1 = src[1];
    const unsigned cw[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
    int4_ Ah, Al;
    {
      int h[4], l[4];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        h[k] = (int)__builtin_amdgcn_perm(cw[2 * k + 1], cw[2 * k], 0x07050301u);
        l[k] = (int)(__builtin_amdgcn_perm(cw[2 * k + 1], cw[2 * k], 0x06040200u) ^ 0x80808080u);
      }
      Ah = int4_{h[0], h[1], h[2], h[3]}; Al = int4_{l[0], l[1], l[2], l[3]};
    }
    // the destination rows this lane will write (even lanes: rows v = 0, 2, ..; odd lanes: v = 1, 3, ..) are requested now
    uint16_t* dbase = plane + (size_t)((blk / BLK_X) * 32) * PITCH + (blk % BLK_X) * 32 + (i & ~1);
    unsigned pw[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const int v = 2 * k + (lane & 1), row = (v & 3) + 8 * (v >> 2) + 4 * g;
      pw[k] = *(const unsigned*)(dbase + (size_t)row * PITCH);
    }
    int16_ zero, lo0;
#pragma unroll
    for (int v = 0; v < 16; v++) { zero[v] = 0; lo0[v] = 128 * cs + 64; }
    int16_ hi = __builtin_amdgcn_mfma_i32_32x32x32_i8(Ah, B1, zero, 0, 0, 0);
    int16_ lo = __builtin_amdgcn_mfma_i32_32x32x32_i8(Al, B1, lo0, 0, 0, 0);
    // first-stage values: clip16(((hi << 8) + lo) >> 7), packed (v, v + 1) with the saturating pack, then split into bytes again
    unsigned gpk[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const int a = (hi[2 * k] * 256 + lo[2 * k]) >> 7, b = (hi[2 * k + 1] * 256 + lo[2 * k + 1]) >> 7;
      gpk[k] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pk_i16(a, b));
    }
    int4_ Gh, Gl;
    {
      int h[4], l[4];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        h[k] = (int)__builtin_amdgcn_perm(gpk[2 * k + 1], gpk[2 * k], 0x07050301u);
        l[k] = (int)(__builtin_amdgcn_perm(gpk[2 * k + 1], gpk[2 * k], 0x06040200u) ^ 0x80808080u);
      }
      Gh = int4_{h[0], h[1], h[2], h[3]}; Gl = int4_{l[0], l[1], l[2], l[3]};
    }
#pragma unroll
    for (int v = 0; v < 16; v++) lo0[v] = 128 * cs + (1 << (post - 1));
    hi = __builtin_amdgcn_mfma_i32_32x32x32_i8(Gh, B2, zero, 0, 0, 0);
    lo = __builtin_amdgcn_mfma_i32_32x32x32_i8(Gl, B2, lo0, 0, 0, 0);
    // residual rows: lane holds x = lane & 31 of rows (v & 3) + 8 (v >> 2) + 4 g; pairs (v, v + 1) are packed (saturating: beyond int16 the sum
    // is clipped either way), lanes x and x ^ 1 swap halves so that each owns two adjacent samples of ONE row, then packed add + clip
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const int a = (hi[2 * k] * 256 + lo[2 * k]) >> post, b = (hi[2 * k + 1] * 256 + lo[2 * k + 1]) >> post;
      const unsigned own = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pk_i16(a, b));
      const unsigned nbr = (unsigned)__builtin_amdgcn_mov_dpp((int)own, 0xB1, 0xF, 0xF, true);      // quad_perm [1,0,3,2]
      const unsigned r = (lane & 1) ? __builtin_amdgcn_perm(own, nbr, sel) : __builtin_amdgcn_perm(nbr, own, sel);
      typedef short s2 __attribute__((ext_vector_type(2)));
      s2 t = __builtin_elementwise_add_sat(__builtin_bit_cast(s2, pw[k]), __builtin_bit_cast(s2, r));
      t = __builtin_elementwise_min(__builtin_elementwise_max(t, (s2)(short)0), __builtin_bit_cast(s2, maxv));
      const int v = 2 * k + (lane & 1), row = (v & 3) + 8 * (v >> 2) + 4 * g;
      *(unsigned*)(dbase + (size_t)row * PITCH) = __builtin_bit_cast(unsigned, t);
    }
  }
}

// ---- host reference (fallback-dct.cc:550-691 for nT = 32) ----
static void ref_block(const int16_t* c, uint16_t* d, int pitch, int bd)
{
  int16_t g[32 * 32];
  for (int col = 0; col < 32; col++)
    for (int i = 0; i < 32; i++) {
      int s = 0;
      for (int j = 0; j < 32; j++) s += M32(j, i) * c[j * 32 + col];
      s = (s + 64) >> 7;
      g[i * 32 + col] = (int16_t)(s < -32768 ? -32768 : (s > 32767 ? 32767 : s));
    }
  const int post = 20 - bd, maxv = (1 << bd) - 1;
  for (int y = 0; y < 32; y++)
    for (int i = 0; i < 32; i++) {
      int s = 0;
      for (int j = 0; j < 32; j++) s += M32(j, i) * g[y * 32 + j];
      const int out = (s + (1 << (post - 1))) >> post;
      const int v = d[y * pitch + i] + out;
      d[y * pitch + i] = (uint16_t)(v < 0 ? 0 : (v > maxv ? maxv : v));
    }
}

int main(int argc, char** argv)
{
  const int nblk = argc > 1 ? atoi(argv[1]) : 16384, bd = 10, reps = 20;
  const int rows = (nblk + BLK_X - 1) / BLK_X * 32;
  std::vector<uint16_t> plane0((size_t)rows * PITCH);
  uint32_t s = 0xC5C5C5C5u;
  auto rnd = [&]() { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return s; };
  for (auto& v : plane0) v = (uint16_t)(rnd() & 1023);
  int16_t *d_ct; uint32_t* d_pairs; uint16_t* d_plane;
  CHK(hipMalloc(&d_ct, (size_t)nblk * 2048)); CHK(hipMalloc(&d_pairs, (size_t)nblk * 2048)); CHK(hipMalloc(&d_plane, plane0.size() * 2));
  hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  int* d_tab;
  {
    std::vector<int> tab(64 * 12, 0);
    for (int lane = 0; lane < 64; lane++) {
      const int i = lane & 31, g = lane >> 5;
      for (int s2 = 0; s2 < 16; s2++) {
        tab[lane * 12 + (s2 >> 2)] |= (int)((unsigned)(M32(16 * g + s2, i) & 0xFF) << (8 * (s2 & 3)));
        const int col = (s2 & 3) + 8 * (s2 >> 2) + 4 * g;
        tab[lane * 12 + 4 + (s2 >> 2)] |= (int)((unsigned)(M32(col, i) & 0xFF) << (8 * (s2 & 3)));
      }
      for (int j = 0; j < 32; j++) tab[lane * 12 + 8] += M32(j, i);
    }
    CHK(hipMalloc(&d_tab, tab.size() * 4));
    CHK(hipMemcpy(d_tab, tab.data(), tab.size() * 4, hipMemcpyHostToDevice));
  }
  printf("# %d blocks of 32x32, %d-bit plane; ns per block = kernel time / blocks (mean of %d launches)\n", nblk, bd, reps);
  for (int occ : {4, 8, 16, 32}) {
    // coefficients: the three scenarios of dev-tools/test-transform.cc (sparse small / dense +-2048 / full int16) inside the top-left occ x occ corner
    std::vector<int16_t> coef((size_t)nblk * 1024, 0), ct((size_t)nblk * 1024);
    std::vector<uint32_t> pairs((size_t)nblk * 512);
    for (int b = 0; b < nblk; b++) {
      const int kind = rnd() % 20;
      for (int j = 0; j < occ; j++)
        for (int c = 0; c < occ; c++) {
          int v;
          if (kind < 16) v = (rnd() % 8 == 0) ? (int)(rnd() % 1025) - 512 : 0;
          else if (kind < 19) v = (int)(rnd() % 4097) - 2048;
          else v = (int)(int16_t)rnd();
          coef[(size_t)b * 1024 + j * 32 + c] = (int16_t)v;
        }
      if (b == 0) for (int k = 0; k < occ * 32; k++) coef[(k / occ) % occ * 32 + k % occ] = (k & 1) ? 32767 : -32768;   // extremes
      for (int j = 0; j < 32; j++)
        for (int c = 0; c < 32; c++) ct[(size_t)b * 1024 + c * 32 + j] = coef[(size_t)b * 1024 + j * 32 + c];
      for (int q = 0; q < 16; q++)
        for (int c = 0; c < 32; c++) pairs[(size_t)b * 512 + q * 32 + c] = (uint16_t)coef[(size_t)b * 1024 + 2 * q * 32 + c] | ((uint32_t)(uint16_t)coef[(size_t)b * 1024 + (2 * q + 1) * 32 + c] << 16);
    }
    std::vector<uint16_t> want = plane0;
    const int ncheck = nblk < 2048 ? nblk : 2048;
    for (int b = 0; b < ncheck; b++) ref_block(&coef[(size_t)b * 1024], &want[(size_t)(b / BLK_X) * 32 * PITCH + (b % BLK_X) * 32], PITCH, bd);
    CHK(hipMemcpy(d_ct, ct.data(), ct.size() * 2, hipMemcpyHostToDevice));
    CHK(hipMemcpy(d_pairs, pairs.data(), pairs.size() * 4, hipMemcpyHostToDevice));
    for (int which = 0; which < 4; which++) {
      const int per_wave = which == 3 ? 16 : (which == 2 ? 4 : 1);
      float tot = 0;
      bool ok = true;
      for (int r = 0; r < reps + 1; r++) {
        CHK(hipMemcpy(d_plane, plane0.data(), plane0.size() * 2, hipMemcpyHostToDevice));
        CHK(hipEventRecord(e0));
        if (which == 0) hipLaunchKernelGGL(k_dot2, dim3((nblk + 1) / 2), dim3(64), 0, 0, d_pairs, d_plane, nblk, occ, bd);
        else hipLaunchKernelGGL(k_mfma, dim3((nblk + 4 * per_wave - 1) / (4 * per_wave)), dim3(256), 0, 0, d_ct, d_plane, nblk, bd, per_wave, d_tab);
        CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
        if (r) tot += ms;
        if (r == 0) {
          std::vector<uint16_t> got(plane0.size());
          CHK(hipMemcpy(got.data(), d_plane, got.size() * 2, hipMemcpyDeviceToHost));
          const size_t n = (size_t)((ncheck + BLK_X - 1) / BLK_X) * 32 * PITCH;
          size_t bad = 0;
          for (size_t k = 0; k < n; k++) {
            const int b = (int)((k / PITCH / 32) * BLK_X + (k % PITCH) / 32);
            if (b < ncheck && got[k] != want[k]) { if (!bad) printf("   first mismatch: block %d sample (%zu,%zu) got %d want %d\n", b, k / PITCH, k % PITCH, got[k], want[k]); bad++; }
          }
          ok = bad == 0;
        }
      }
      printf("occupied %2dx%-2d  %-22s %8.1f ns/block  (%.4f ms)  %s\n", occ, occ, which == 0 ? "k_dot2 (product scheme)" : (which == 1 ? "k_mfma 1 block/wave" : (which == 2 ? "k_mfma 4 blocks/wave" : "k_mfma 16 blocks/wave")),
             tot / reps * 1e6 / nblk, tot / reps, ok ? "bit-exact" : "MISMATCH");
    }
  }
  return 0;
}
