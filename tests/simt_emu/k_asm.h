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
static inline int d_dot2(unsigned a, unsigned b, int c)
{
  return c + (int)(int16_t)(a & 0xFFFF) * (int16_t)(b & 0xFFFF) + (int)(int16_t)(a >> 16) * (int16_t)(b >> 16);
}
#endif
