/* tests/simt_emu/k_asm.h — interpreter stand-in for libde265_amd/csrc/k_asm.h (test tier only). */
#ifndef M355_K_ASM_H
#define M355_K_ASM_H
#include <string.h>
#define M355_GLOBAL
/* workgroups run one after the other here: a poll that is not satisfied at once never will be */
static inline void d_touch(const void*, unsigned*) {}
static inline void d_st_nt4(void* p, unsigned v) { *(unsigned*)p = v; }
static inline void d_st_nt8(void* p, unsigned v0, unsigned v1) { ((unsigned*)p)[0] = v0; ((unsigned*)p)[1] = v1; }
#define M355_PIN_S(x) ((void)0)
#define M355_PIN_V(x) ((void)0)
#define M355_COMPILER_FENCE() ((void)0)
#define M355_PIN_V4_MEM(a, b, c, d) ((void)0)

#define M355_SPIN_LIMIT 4u
static inline void d_drain_vmem() {}
static inline void d_drain_lds() {}
static inline void d_ldg16(const void* p, unsigned* o) { memcpy(o, p, 16); }
static inline void d_ldg12(const void* p, unsigned* o) { memcpy(o, p, 12); }
static inline void d_ldg8(const void* p, unsigned* o) { memcpy(o, p, 8); }
static inline unsigned d_ldg4(const void* p) { unsigned v; memcpy(&v, p, 4); return v; }
static inline unsigned d_ldg2(const void* p) { unsigned short v; memcpy(&v, p, 2); return v; }
static inline void d_stg16(void* p, const unsigned* v) { memcpy(p, v, 16); }
static inline void d_stg8(void* p, const unsigned* v) { memcpy(p, v, 8); }
static inline void d_stg4(void* p, unsigned v) { memcpy(p, &v, 4); }
static inline unsigned d_pack_lo16(unsigned lo, unsigned hi) { return (lo & 0xFFFFu) | (hi << 16); }
static inline unsigned d_pack_hi16(unsigned lo, unsigned hi) { return (lo >> 16) | (hi & 0xFFFF0000u); }
static inline unsigned d_pk_shl16(unsigned v, int s) { return ((v << s) & 0xFFFFu) | ((((v >> 16) << s) & 0xFFFFu) << 16); }
static inline unsigned d_byte_lookup(unsigned hi, unsigned lo, unsigned idx) { const unsigned long long t = ((unsigned long long)hi << 32) | lo; return (unsigned)(t >> (8 * (idx & 7))) & 0xFFu; }
#define M355_PK2(expr_lo, expr_hi) ((unsigned)((expr_lo) & 0xFFFFu) | ((unsigned)((expr_hi) & 0xFFFFu) << 16))
static inline unsigned d_pk_sub16(unsigned a, unsigned b) { return M355_PK2((a & 0xFFFF) - (b & 0xFFFF), (a >> 16) - (b >> 16)); }
static inline unsigned d_pk_add16(unsigned a, unsigned b) { return M355_PK2((a & 0xFFFF) + (b & 0xFFFF), (a >> 16) + (b >> 16)); }
static inline unsigned d_pk_lshr16(unsigned v, int s) { return M355_PK2((v & 0xFFFF) >> s, (v >> 16) >> s); }
static inline int m355_s16(unsigned v) { return (int)(short)(v & 0xFFFF); }
static inline unsigned d_pk_min_i16(unsigned a, unsigned b) { const int l = m355_s16(a) < m355_s16(b) ? m355_s16(a) : m355_s16(b), h = m355_s16(a >> 16) < m355_s16(b >> 16) ? m355_s16(a >> 16) : m355_s16(b >> 16); return M355_PK2((unsigned)l, (unsigned)h); }
static inline unsigned d_pk_max_i16(unsigned a, unsigned b) { const int l = m355_s16(a) > m355_s16(b) ? m355_s16(a) : m355_s16(b), h = m355_s16(a >> 16) > m355_s16(b >> 16) ? m355_s16(a >> 16) : m355_s16(b >> 16); return M355_PK2((unsigned)l, (unsigned)h); }
static inline unsigned d_pk_min_u16(unsigned a, unsigned b) { const unsigned l = (a & 0xFFFF) < (b & 0xFFFF) ? (a & 0xFFFF) : (b & 0xFFFF), h = (a >> 16) < (b >> 16) ? (a >> 16) : (b >> 16); return M355_PK2(l, h); }
static inline unsigned d_pk_addsat_u16(unsigned a, unsigned b) { unsigned l = (a & 0xFFFF) + (b & 0xFFFF), h = (a >> 16) + (b >> 16); if (l > 0xFFFF) l = 0xFFFF; if (h > 0xFFFF) h = 0xFFFF; return M355_PK2(l, h); }
static inline unsigned d_pk_subsat_u16(unsigned a, unsigned b) { const unsigned l = (a & 0xFFFF) > (b & 0xFFFF) ? (a & 0xFFFF) - (b & 0xFFFF) : 0, h = (a >> 16) > (b >> 16) ? (a >> 16) - (b >> 16) : 0; return M355_PK2(l, h); }
static inline unsigned d_perm(unsigned hi, unsigned lo, unsigned sel)
{
  const unsigned long long t = ((unsigned long long)hi << 32) | lo;
  unsigned r = 0;
  for (int i = 0; i < 4; i++) {
    const unsigned s = (sel >> (8 * i)) & 0xFF;
    const unsigned b = s < 8 ? (unsigned)(t >> (8 * s)) & 0xFF : (s == 0x0c ? 0u : 0xFFu);
    r |= b << (8 * i);
  }
  return r;
}
static inline int d_dot2(unsigned a, unsigned b, int c)
{
  return c + (int)(int16_t)(a & 0xFFFF) * (int16_t)(b & 0xFFFF) + (int)(int16_t)(a >> 16) * (int16_t)(b >> 16);
}
static inline int d_dot2z(unsigned a, unsigned b) { return d_dot2(a, b, 0); }
static inline int d_dot4(unsigned a, unsigned b, int c)
{
  for (int i = 0; i < 4; i++) c += (int)(signed char)(a >> (8 * i)) * (int)(signed char)(b >> (8 * i));
  return c;
}
static inline int d_dot4k(unsigned a, unsigned b, int k) { return d_dot4(a, b, k); }
static inline unsigned d_pack_mid16(unsigned lo, unsigned hi) { return ((lo >> 8) & 0xFFFFu) | ((hi >> 8) << 16); }
static inline unsigned d_pack_bytes(unsigned lo, unsigned hi) { return (lo & 0xFFu) | ((lo >> 8) & 0xFF00u) | ((hi & 0xFFu) << 16) | ((hi & 0xFF0000u) << 8); }
static inline int m355_sat16(int v) { return v < -32768 ? -32768 : (v > 32767 ? 32767 : v); }
static inline unsigned d_pk_addsat_i16(unsigned a, unsigned b) { return M355_PK2((unsigned)m355_sat16(m355_s16(a) + m355_s16(b)), (unsigned)m355_sat16(m355_s16(a >> 16) + m355_s16(b >> 16))); }
static inline unsigned d_pk_ashr16(unsigned v, int s) { return M355_PK2((unsigned)(m355_s16(v) >> s), (unsigned)(m355_s16(v >> 16) >> s)); }
#endif
