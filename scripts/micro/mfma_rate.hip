// mfma_rate.hip — diagnostic: sustained v_mfma_f32_32x32x16_bf16 rate of the chip, operands in registers or one
// ds_read_b128 per MFMA, 1-3 waves per SIMD, and the s_memtime tick.   hipcc --offload-arch=gfx950 -O3 mfma_rate.hip -o mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ void __launch_bounds__(256) k(float *out, unsigned long long *ticks, int iters) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 16 * 1024 / 4; i += blockDim.x) lds[i] = (float)i * 1e-9f;
  __syncthreads();
  f32x16 acc0 = {}, acc1 = {};
  bf16x8 a = __builtin_bit_cast(bf16x8, make_float4(1.f + lane, 2.f, 3.f, 4.f)), b = __builtin_bit_cast(bf16x8, make_float4(0.5f, lane, 1.5f, 2.5f));
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      if (MODE == 1) b = __builtin_bit_cast(bf16x8, *reinterpret_cast<const float4 *>(lds + ((c * 64 + lane) * 4 + (it & 1) * 2048)));
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      if (MODE == 1) b = __builtin_bit_cast(bf16x8, *reinterpret_cast<const float4 *>(lds + ((c * 64 + lane) * 4 + 1024)));
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

int main() {
  int cus = 0;
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  float *out; unsigned long long *ticks;
  hipMalloc(&out, 256 * 3 * cus * 4 * 8); hipMalloc(&ticks, 8 * 3 * cus * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 2; ++mode)
    for (int wgs = 1; wgs <= 3; ++wgs)
      for (int iters : {2000, 20000}) {
        const int grid = wgs * cus;
        for (int rep = 0; rep < 2; ++rep) {
          hipEventRecord(e0);
          if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 16384, 0, out, ticks, iters);
          else hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 16384, 0, out, ticks, iters);
          hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(grid);
        hipMemcpy(h.data(), ticks, grid * 8, hipMemcpyDeviceToHost);
        const double mf = (double)grid * 4 * iters * 16;  // MFMAs
        printf("mode %d (%s) waves/SIMD %d iters %6d: %8.3f ms  %7.1f TFLOP/s  cycles/MFMA/SIMD at 2.4 GHz %.1f  ticks %llu -> %.2f ns/tick\n", mode,
               mode ? "ds_read_b128 per MFMA" : "register operands", wgs, iters, ms, mf * 32768 / (ms * 1e-3) / 1e12,
               (ms * 1e-3) * 2.4e9 / (iters * 16.0 * wgs), h[0], ms * 1e6 / (double)h[0]);
      }
  return 0;
}
