/* tests/simt_emu/k_asm.h — interpreter stand-in for libde265_amd/csrc/k_asm.h (test tier only). */
#ifndef M355_K_ASM_H
#define M355_K_ASM_H
static inline void d_drain_vmem() {}
#endif
