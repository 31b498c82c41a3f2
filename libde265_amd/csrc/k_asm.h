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

#endif
