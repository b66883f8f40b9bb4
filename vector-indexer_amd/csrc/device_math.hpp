// device_math.hpp — exact-order squared-L2 device functions (must be compiled with
// -ffp-contract=off; a fused multiply-add changes the rounding of acc + t*t).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/vi_reduce_order.h"

namespace vi {

__device__ __forceinline__ void sq_add(float &acc, float q, float x) {
  const float t = q - x;
  acc = acc + t * t;
}

// euclidean_distance_squared — src/utils.rs:28-30
__device__ __forceinline__ float l2sq_scalar_dev(const float *p, const float *c, uint32_t d) {
  float acc = 0.0f;
  for (uint32_t j = 0; j < d; ++j) sq_add(acc, p[j], c[j]);
  return acc;
}

// compute_distance_simd — src/kmeans.rs:377-419 (wide 0.7.33 reduce order: see oracle/vi_oracle.h)
__device__ __forceinline__ float l2sq_lanes_dev(const float *p, const float *c, uint32_t d) {
  float a8[8] = {0, 0, 0, 0, 0, 0, 0, 0}, a4[4] = {0, 0, 0, 0}, tail = 0.0f;
  uint32_t j = 0;
  for (; j + 8 <= d; j += 8)
#pragma unroll
    for (int l = 0; l < 8; ++l) sq_add(a8[l], p[j + l], c[j + l]);
  for (; j + 4 <= d; j += 4)
#pragma unroll
    for (int l = 0; l < 4; ++l) sq_add(a4[l], p[j + l], c[j + l]);
  for (; j < d; ++j) sq_add(tail, p[j], c[j]);
  const float lo = VI_REDUCE4(a8[0], a8[1], a8[2], a8[3]);  // lane order: include/vi_reduce_order.h (unpinned)
  const float hi = VI_REDUCE4(a8[4], a8[5], a8[6], a8[7]);
  const float r4 = VI_REDUCE4(a4[0], a4[1], a4[2], a4[3]);
  return ((lo + hi) + r4) + tail;
}

}  // namespace vi
