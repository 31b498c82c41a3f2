/* tests/simt_emu/k_asm.h — interpreter stand-in for libde265_amd/csrc/k_asm.h (test tier only). */
#ifndef M355_K_ASM_H
#define M355_K_ASM_H
#include <string.h>
#define M355_GLOBAL
static inline void d_drain_vmem() {}
static inline void d_ldg16(const void* p, unsigned* o) { memcpy(o, p, 16); }
static inline void d_ldg12(const void* p, unsigned* o) { memcpy(o, p, 12); }
static inline void d_ldg8(const void* p, unsigned* o) { memcpy(o, p, 8); }
static inline unsigned d_ldg4(const void* p) { unsigned v; memcpy(&v, p, 4); return v; }
static inline unsigned d_ldg2(const void* p) { unsigned short v; memcpy(&v, p, 2); return v; }
static inline unsigned d_pack_lo16(unsigned lo, unsigned hi) { return (lo & 0xFFFFu) | (hi << 16); }
static inline unsigned d_pack_hi16(unsigned lo, unsigned hi) { return (lo >> 16) | (hi & 0xFFFF0000u); }
static inline unsigned d_pk_shl16(unsigned v, int s) { return ((v << s) & 0xFFFFu) | ((((v >> 16) << s) & 0xFFFFu) << 16); }
static inline unsigned d_byte_lookup(unsigned hi, unsigned lo, unsigned idx) { const unsigned long long t = ((unsigned long long)hi << 32) | lo; return (unsigned)(t >> (8 * (idx & 7))) & 0xFFu; }
static inline int d_dot2(unsigned a, unsigned b, int c)
{
  return c + (int)(int16_t)(a & 0xFFFF) * (int16_t)(b & 0xFFFF) + (int)(int16_t)(a >> 16) * (int16_t)(b >> 16);
}
#endif
