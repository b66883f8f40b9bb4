// Micro-benchmark: f32 VALU issue rate on gfx950 for the scan kernel's op mix (sub, mul, add),
// scalar vs packed, as a function of waves per SIMD.  Diagnostic only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float float2_t __attribute__((ext_vector_type(2)));

template <int CH>
__global__ void __launch_bounds__(256) k_scalar(float *out, float q, int iters) {
  float acc[CH], x[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) { acc[c] = 0.f; x[c] = threadIdx.x * 0.001f + c; }
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      float t = q - x[c];
      acc[c] = acc[c] + t * t;
      x[c] = t;
    }
  }
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < CH; ++c) s += acc[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int CH>
__global__ void __launch_bounds__(256) k_packed(float *out, float q, int iters) {
  float2_t acc[CH], x[CH];
  const float2_t qq = {q, q + 1.f};
#pragma unroll
  for (int c = 0; c < CH; ++c) { acc[c] = (float2_t){0.f, 0.f}; x[c] = (float2_t){threadIdx.x * 0.001f + c, threadIdx.x * 0.002f + c}; }
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      float2_t t = qq - x[c];
      float2_t m = t * t;
      acc[c] = acc[c] + m;
      x[c] = t;
    }
  }
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < CH; ++c) s += acc[c].x + acc[c].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
double time_ms(F launch) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  launch();
  hipDeviceSynchronize();
  hipEventRecord(a);
  launch();
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  return ms;
}

int main() {
  float *out;
  hipMalloc(&out, 256 * 256 * 64 * sizeof(float));
  const int iters = 20000;
  for (int wps : {1, 2, 3, 4, 8}) {           // waves per SIMD
    const int blocks = 256 * wps;             // 256 CUs, block = 4 waves = 1 wave per SIMD
    {
      double ms = time_ms([&] { hipLaunchKernelGGL(k_scalar<4>, dim3(blocks), dim3(256), 0, 0, out, 0.5f, iters); });
      double ops = (double)blocks * 256 * iters * 4 * 3;  // lane-ops
      printf("scalar CH=4 wps=%d  %.3f ms  %.2f T lane-ops/s  (%.2f cyc/instr/SIMD @2.4GHz)\n", wps, ms,
             ops / ms / 1e9, ms * 1e-3 * 2.4e9 / ((double)iters * 4 * 3 * wps));
    }
    {
      double ms = time_ms([&] { hipLaunchKernelGGL(k_scalar<8>, dim3(blocks), dim3(256), 0, 0, out, 0.5f, iters); });
      double ops = (double)blocks * 256 * iters * 8 * 3;
      printf("scalar CH=8 wps=%d  %.3f ms  %.2f T lane-ops/s  (%.2f cyc/instr/SIMD)\n", wps, ms, ops / ms / 1e9,
             ms * 1e-3 * 2.4e9 / ((double)iters * 8 * 3 * wps));
    }
    {
      double ms = time_ms([&] { hipLaunchKernelGGL(k_packed<4>, dim3(blocks), dim3(256), 0, 0, out, 0.5f, iters); });
      double ops = (double)blocks * 256 * iters * 4 * 3 * 2;
      printf("packed CH=4 wps=%d  %.3f ms  %.2f T lane-ops/s  (%.2f cyc/pk-instr/SIMD)\n", wps, ms, ops / ms / 1e9,
             ms * 1e-3 * 2.4e9 / ((double)iters * 4 * 3 * wps));
    }
  }
  return 0;
}
