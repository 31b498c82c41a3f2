/*
 * k_asm.h — the few gfx950 instruction-level idioms the kernels need, kept in one place.
 * (tests/simt_emu/ shadows this header with interpreter equivalents; the product always uses this.)
 */
#ifndef M355_K_ASM_H
#define M355_K_ASM_H

/* Drain this wave's outstanding vector-memory operations. Written as inline asm on purpose: after
 * an agent-scope release fence hipcc may drop a builtin s_waitcnt whose counter it believes to be
 * zero, letting the flag store overtake the L2 write-back (MI355X guide, "Compiler hazard"). */
__device__ __forceinline__ void d_drain_vmem() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
/* this wave's LDS operations have completed (and nothing else is waited for: vector loads stay in flight) */
__device__ __forceinline__ void d_drain_lds() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

/* streaming stores (the `nt` bit): written samples do not displace the reference windows other workgroups are reading through the
 * same L2 — k_inter_jobs' own output is next read a kernel later, when the 4 MB L2 of an XCD has long been turned over */
__device__ __forceinline__ void d_st_nt4(void* p, unsigned v) { __builtin_nontemporal_store(v, (unsigned*)p); }
__device__ __forceinline__ void d_st_nt8(void* p, unsigned v0, unsigned v1) { __builtin_nontemporal_store(v0, (unsigned*)p); __builtin_nontemporal_store(v1, (unsigned*)p + 1); }

/* no memory operation moves across this point (compiler only; costs no instruction) */
#define M355_COMPILER_FENCE() asm volatile("" ::: "memory")

/* keep a wave-uniform value in a scalar register, computed HERE (k_intra decodes its next block's record before the level barrier,
 * not at the first use behind it) */
#define M355_PIN_S(x) asm volatile("" : "+s"(x))
/* make a value opaque to the optimiser where it stands (any register): e.g. two record fields that must both be LOADED before a select
 * between them — hipcc otherwise selects the ADDRESS and loads one of them later, behind whatever is in flight */
#define M355_PIN_V(x) asm volatile("" : "+v"(x))

/* four results pinned where they stand + a compiler memory barrier: the arithmetic that makes them cannot sink below this point and no
 * load behind it can be hoisted above it (k_inter_jobs' software pipeline) */
#define M355_PIN_V4_MEM(a, b, c, d) asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "memory")

/* spin bound of k_intra's granule polls: ~2^22 polls x (one L2 round trip + s_sleep) is seconds — far beyond any real wait */
#define M355_SPIN_LIMIT (1u << 22)

/* pointers loaded from device tables: tell hipcc they are global (else it emits flat_load) */
#define M355_GLOBAL __attribute__((address_space(1)))

/* request a cache line without using it: a later load of the same line then finds it in the L2 instead of paying a whole HBM
 * round trip on a dependent chain (k_intra: the residuals of a CTB's blocks, consumed one dependency level at a time).  The
 * data goes to a 256-byte LDS scratch area (global_load_lds_dword: lane i writes scratch[i]) that nobody reads: no VGPR is
 * tied up and nothing ever waits for it. */
__device__ __forceinline__ void d_touch(const void* p, unsigned* lds_scratch64)
{
  __builtin_amdgcn_global_load_lds((const M355_GLOBAL void*)p, (__attribute__((address_space(3))) void*)lds_scratch64, 4, 0, 0);
}


/* unaligned (2-byte / 1-byte aligned) vector loads from global memory: gfx950 executes them as single
 * global_load_dwordx4/x3/x2 (checked on hardware, tools/ubench/ub_inter.hip) */
typedef unsigned m355_u4 __attribute__((ext_vector_type(4), aligned(1)));
typedef unsigned m355_u3 __attribute__((ext_vector_type(3), aligned(1)));
typedef unsigned m355_u2 __attribute__((ext_vector_type(2), aligned(1)));
typedef unsigned m355_u1 __attribute__((aligned(1)));
typedef unsigned short m355_h1 __attribute__((aligned(1)));
__device__ __forceinline__ void d_ldg16(const M355_GLOBAL void* p, unsigned* o) { const m355_u4 v = *(const M355_GLOBAL m355_u4*)p; o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; }
__device__ __forceinline__ void d_ldg12(const M355_GLOBAL void* p, unsigned* o) { const m355_u3 v = *(const M355_GLOBAL m355_u3*)p; o[0] = v.x; o[1] = v.y; o[2] = v.z; }
__device__ __forceinline__ void d_ldg8(const M355_GLOBAL void* p, unsigned* o) { const m355_u2 v = *(const M355_GLOBAL m355_u2*)p; o[0] = v.x; o[1] = v.y; }
__device__ __forceinline__ unsigned d_ldg4(const M355_GLOBAL void* p) { return *(const M355_GLOBAL m355_u1*)p; }
__device__ __forceinline__ unsigned d_ldg2(const M355_GLOBAL void* p) { return *(const M355_GLOBAL m355_h1*)p; }
/* the matching stores (plain: the line stays in the L2 for whoever reads the samples next) */
__device__ __forceinline__ void d_stg16(M355_GLOBAL void* p, const unsigned* v) { m355_u4 t; t.x = v[0]; t.y = v[1]; t.z = v[2]; t.w = v[3]; *(M355_GLOBAL m355_u4*)p = t; }
__device__ __forceinline__ void d_stg8(M355_GLOBAL void* p, const unsigned* v) { m355_u2 t; t.x = v[0]; t.y = v[1]; *(M355_GLOBAL m355_u2*)p = t; }
__device__ __forceinline__ void d_stg4(M355_GLOBAL void* p, unsigned v) { *(M355_GLOBAL m355_u1*)p = v; }

/* v_perm_b32: byte permute of {hi:lo}; the two selectors used here gather the LOW resp. HIGH 16-bit halves
 * of two registers into one packed pair (lo -> bits 0..15, hi -> bits 16..31) in a single VALU issue */
__device__ __forceinline__ unsigned d_pack_lo16(unsigned lo, unsigned hi) { return __builtin_amdgcn_perm(hi, lo, 0x05040100u); }
__device__ __forceinline__ unsigned d_pack_hi16(unsigned lo, unsigned hi) { return __builtin_amdgcn_perm(hi, lo, 0x07060302u); }
/* v_perm_b32 as an 8-entry byte table lookup: byte idx (0..7) of {hi:lo} */
__device__ __forceinline__ unsigned d_byte_lookup(unsigned hi, unsigned lo, unsigned idx) { return __builtin_amdgcn_perm(hi, lo, idx) & 0xFFu; }
/* v_pk_lshlrev_b16: both 16-bit halves shifted left by s */
typedef unsigned short m355_ushort2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned d_pk_shl16(unsigned v, int s)
{
  m355_ushort2 t = __builtin_bit_cast(m355_ushort2, v);
  t = t << (m355_ushort2)(unsigned short)s;
  return __builtin_bit_cast(unsigned, t);
}

/* packed 16-bit lane arithmetic (VOP3P: two samples per VALU issue) — k_sao's edge / band classification */
__device__ __forceinline__ unsigned d_pk_sub16(unsigned a, unsigned b) { return __builtin_bit_cast(unsigned, __builtin_bit_cast(m355_ushort2, a) - __builtin_bit_cast(m355_ushort2, b)); }
__device__ __forceinline__ unsigned d_pk_add16(unsigned a, unsigned b) { return __builtin_bit_cast(unsigned, __builtin_bit_cast(m355_ushort2, a) + __builtin_bit_cast(m355_ushort2, b)); }
__device__ __forceinline__ unsigned d_pk_lshr16(unsigned v, int s) { return __builtin_bit_cast(unsigned, __builtin_bit_cast(m355_ushort2, v) >> (m355_ushort2)(unsigned short)s); }
typedef short m355_short2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned d_pk_min_i16(unsigned a, unsigned b) { return __builtin_bit_cast(unsigned, __builtin_elementwise_min(__builtin_bit_cast(m355_short2v, a), __builtin_bit_cast(m355_short2v, b))); }
__device__ __forceinline__ unsigned d_pk_max_i16(unsigned a, unsigned b) { return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(m355_short2v, a), __builtin_bit_cast(m355_short2v, b))); }
__device__ __forceinline__ unsigned d_pk_min_u16(unsigned a, unsigned b) { return __builtin_bit_cast(unsigned, __builtin_elementwise_min(__builtin_bit_cast(m355_ushort2, a), __builtin_bit_cast(m355_ushort2, b))); }
__device__ __forceinline__ unsigned d_pk_addsat_u16(unsigned a, unsigned b) { return __builtin_bit_cast(unsigned, __builtin_elementwise_add_sat(__builtin_bit_cast(m355_ushort2, a), __builtin_bit_cast(m355_ushort2, b))); }
__device__ __forceinline__ unsigned d_pk_subsat_u16(unsigned a, unsigned b) { return __builtin_bit_cast(unsigned, __builtin_elementwise_sub_sat(__builtin_bit_cast(m355_ushort2, a), __builtin_bit_cast(m355_ushort2, b))); }
/* raw v_perm_b32: result byte i = byte sel[i] of {hi:lo} for selectors 0..7, 0x00 for 0x0c, 0xff for 0x0d..0x0f */
__device__ __forceinline__ unsigned d_perm(unsigned hi, unsigned lo, unsigned sel) { return __builtin_amdgcn_perm(hi, lo, sel); }

/* v_dot2c_i32_i16: c + a.lo*b.lo + a.hi*b.hi on packed signed 16-bit pairs — two filter taps per VALU
 * issue (measured on MI355X: same issue rate as v_mad_i32_i24, tools/ubench/ub_inter.hip). */
typedef short m355_short2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int d_dot2(unsigned a, unsigned b, int c)
{
  return __builtin_amdgcn_sdot2(__builtin_bit_cast(m355_short2, a), __builtin_bit_cast(m355_short2, b), c, false);
}

/* head of a dot2 chain: v_dot2_i32_i16 (VOP3P) with the inline constant 0 as its accumulator.  Through the builtin hipcc picks the two-operand
 * v_dot2c (accumulates in place) and zeroes the destination with a v_mov in front of every chain — one extra issue per four or five dot2 */
__device__ __forceinline__ int d_dot2z(unsigned a, unsigned b)
{
  int r;
  asm("v_dot2_i32_i16 %0, %1, %2, 0" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

/* v_dot4_i32_i8: c + sum of four signed-byte products — four filter taps per VALU issue (8-bit planes: samples XOR 0x80 are the
 * signed operand, the +128 * sum(taps) correction sits in the accumulator's start value) */
__device__ __forceinline__ int d_dot4(unsigned a, unsigned b, int c) { return __builtin_amdgcn_sdot4((int)a, (int)b, c, false); }
/* head of a dot4 chain that starts from a constant (the +128 * 64 of the XOR-0x80 operands): the three-operand v_dot4_i32_i8 with the constant in a
 * scalar register instead of v_mov + v_dot4c */
__device__ __forceinline__ int d_dot4k(unsigned a, unsigned b, int k)
{
  int r;
  asm("v_dot4_i32_i8 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(k));
  return r;
}
/* v_perm_b32: bytes 1..2 of two registers -> one packed pair = ((lo >> 8) & 0xFFFF) | ((hi >> 8) << 16): a filter sum whose taps were
 * scaled so that its final right shift is 8 is shifted, truncated to int16 and packed with its neighbour in ONE issue */
__device__ __forceinline__ unsigned d_pack_mid16(unsigned lo, unsigned hi) { return __builtin_amdgcn_perm(hi, lo, 0x06050201u); }
/* bytes 0 and 2 of two registers (the low bytes of four packed 16-bit values) -> four bytes */
__device__ __forceinline__ unsigned d_pack_bytes(unsigned lo, unsigned hi) { return __builtin_amdgcn_perm(hi, lo, 0x06040200u); }
/* packed signed 16-bit: saturating add (v_pk_add_i16 clamp), arithmetic shift right (v_pk_ashrrev_i16) */
__device__ __forceinline__ unsigned d_pk_addsat_i16(unsigned a, unsigned b) { return __builtin_bit_cast(unsigned, __builtin_elementwise_add_sat(__builtin_bit_cast(m355_short2v, a), __builtin_bit_cast(m355_short2v, b))); }
__device__ __forceinline__ unsigned d_pk_ashr16(unsigned v, int s) { return __builtin_bit_cast(unsigned, __builtin_bit_cast(m355_short2v, v) >> (m355_short2v)(short)s); }

#endif
