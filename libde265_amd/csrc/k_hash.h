/* k_hash.h — arguments of the picture-hash kernels (k_hash.hip), shared with runtime.hip */
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <hip/hip_runtime.h>

struct HashPlane {
  const uint8_t* base;
  size_t pitch;        /* bytes between rows */
  int row_bytes;       /* message bytes per row: width * bytes per sample */
  int h;
  int bpp;             /* bytes per sample (1 or 2) */
};
struct HashArgs {
  HashPlane pl[3];
  int first[4];        /* first wave of plane 0,1,2 and the total */
  int rows_per_wave;
  uint32_t* out;       /* 3 accumulators, zeroed by the caller on the same stream */
};

void m355_launch_frame_hash(const HashArgs& a, int type, hipStream_t st);
uint32_t m355_crc_init_term(uint64_t nbytes);     /* init * x^(8 nbytes): what the host XORs onto the device accumulator */
void m355_md5_rows(const uint8_t* data, size_t pitch, int row_bytes, int h, uint8_t out[16]);
