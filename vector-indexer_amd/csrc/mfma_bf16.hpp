// mfma_bf16.hpp — bf16 x 3 split arithmetic shared by the MFMA ranking kernels (filter_search.hip,
// assign_mfma.hip).  x = hi + lo + r with hi = bf16(x), lo = bf16(x - hi), |x - hi| <= 2^-8 |x|, |r| <= 2^-17 |x|; a product a*b is
// ranked as a_hi*b_hi + a_hi*b_lo + a_lo*b_hi on v_mfma_f32_32x32x16_bf16 (bf16 pairs multiply exactly in f32).
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

namespace vi {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void *lds_ptr_t;

// minimum of three without the canonicalising v_max the compiler puts in front of fminf on values it cannot prove quiet
__device__ __forceinline__ float min3_raw(float a, float b, float c) {
  float d;
  asm("v_min3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
// the minimum of a 32x32 accumulator tile's 16 registers (12 instructions).  The accumulator registers themselves are
// read by compiler-visible instructions only — v_med3_f32(-inf, x, y) = min(x, y), operands taken as they are: the
// hazard recogniser does not look into inline asm, and an asm v_min3_f32 placed right behind the last MFMA of a chain
// read the accumulator before the matrix pipe had written it (found with the streaming kernel: missed neighbours).
// The -inf comes out of an asm so that the compiler cannot fold the median into canonicalize + v_min (3 instructions).
__device__ __forceinline__ float opaque_neg_inf() {
  float v;
  asm("v_mov_b32 %0, 0xff800000" : "=v"(v));
  return v;
}
__device__ __forceinline__ void tile_min_level1(const f32x16 &a, float ninf, float (&p)[8]) {
#pragma unroll
  for (int i = 0; i < 8; ++i) p[i] = __builtin_amdgcn_fmed3f(ninf, a[2 * i], a[2 * i + 1]);
}
__device__ __forceinline__ float tile_min_level2(const float (&p)[8]) {
  const float m0 = min3_raw(p[0], p[1], p[2]), m1 = min3_raw(p[3], p[4], p[5]), m2 = min3_raw(p[6], p[7], p[7]);
  return min3_raw(m0, m1, m2);
}
__device__ __forceinline__ float tile_min(const f32x16 &a) {
  float p[8];
  tile_min_level1(a, opaque_neg_inf(), p);
  return tile_min_level2(p);
}
// v into the sorted four smallest T0 <= T1 <= T2 <= T3 (v_med3_f32 takes its operands as they are)
#define VI_TOP4(v)                             \
  {                                            \
    T3 = __builtin_amdgcn_fmed3f(T2, T3, v);   \
    T2 = __builtin_amdgcn_fmed3f(T1, T2, v);   \
    T1 = __builtin_amdgcn_fmed3f(T0, T1, v);   \
    T0 = min3_raw(T0, v, v);                   \
  }

__device__ __forceinline__ uint32_t bf16_rn(float x) {  // round-to-nearest-even, finite inputs
  const uint32_t b = __float_as_uint(x);
  return (b + 0x7FFFu + ((b >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ void split_pair(float x0, float x1, uint32_t &hi, uint32_t &lo) {
  const uint32_t h0 = bf16_rn(x0), h1 = bf16_rn(x1);
  const float r0 = x0 - __uint_as_float(h0 << 16), r1 = x1 - __uint_as_float(h1 << 16);  // exact
  hi = h0 | (h1 << 16);
  lo = bf16_rn(r0) | (bf16_rn(r1) << 16);
}
// 8 consecutive values (two float4), scaled, -> packed hi / lo halves (operand of one lane)
__device__ __forceinline__ void split8(const float4 &lo4, const float4 &hi4, float scale, uint4 &hi, uint4 &lo) {
  split_pair(scale * lo4.x, scale * lo4.y, hi.x, lo.x);
  split_pair(scale * lo4.z, scale * lo4.w, hi.y, lo.y);
  split_pair(scale * hi4.x, scale * hi4.y, hi.z, lo.z);
  split_pair(scale * hi4.z, scale * hi4.w, hi.w, lo.w);
}

// Image of 64 vectors for the A operand: [chunk of 16 dims][plane hi/lo][half of 8 dims][64 vectors] x 16 B
// (lane (j, h) reads the 8 consecutive dims 16c+8h.. of vector j with one conflict-free ds_read_b128);
// 2*NG KiB per 64 vectors when the dims are padded to 8*NG, the same bytes as the f32 values.
// tile_dma_image copies one image + the 64 squared norms behind it into LDS with LDS-DMA, 4 waves.
template <int NG, int WAVES = 4>
__device__ __forceinline__ void tile_dma_image(float *tile, const float4 *src, const float *xn, int wave, int lane) {
#pragma unroll
  for (int i = 0; i < 2 * NG / WAVES; ++i) {  // 2*NG pieces of 1 KiB, round-robin over the waves
    const int piece = wave + WAVES * i;
    __builtin_amdgcn_global_load_lds(src + piece * 64 + lane, (lds_ptr_t)(tile + piece * 256), 16, 0, 0);
  }
  if (wave == 0) __builtin_amdgcn_global_load_lds(xn + lane, (lds_ptr_t)(tile + 2 * NG * 256), 4, 0, 0);
}

// The same copies as inline asm.  hipcc does not count asm memory operations, so it inserts no s_waitcnt for them:
// with the builtin it waits vmcnt(0) before the first ds_read that follows an LDS-DMA (it cannot tell that the read
// touches the OTHER buffer), which serialises a double-buffered loop.  Callers place their own counted waits.
// M0 (LDS destination base) is written in the same statement that uses it and restored.
__device__ __forceinline__ void glds16_asm(const void *gsrc, float *lds_dst) {
  unsigned keep;
  const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(size_t)(lds_ptr_t)lds_dst);
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
}
// ... with the destination given as a (wave-uniform) LDS byte address: no generic -> LDS pointer conversion per call
__device__ __forceinline__ void glds16_at(const void *gsrc, unsigned lds_addr) {
  unsigned keep;
  const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_addr);
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
}
__device__ __forceinline__ void glds4_at(const void *gsrc, unsigned lds_addr) {  // 4 bytes per lane
  unsigned keep;
  const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_addr);
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
}
__device__ __forceinline__ void glds4_asm(const void *gsrc, float *lds_dst) {
  unsigned keep;
  const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(size_t)(lds_ptr_t)lds_dst);
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
}

// tile_dma_image with the asm copies (double-buffered loops)
template <int NG, int WAVES = 4>
__device__ __forceinline__ void tile_dma_image_asm(float *tile, const float4 *src, const float *xn, int wave, int lane) {
#pragma unroll
  for (int i = 0; i < 2 * NG / WAVES; ++i) {
    const int piece = wave + WAVES * i;
    glds16_asm(src + piece * 64 + lane, tile + piece * 256);
  }
  if (wave == 0) glds4_asm(xn + lane, tile + 2 * NG * 256);
}

}  // namespace vi
