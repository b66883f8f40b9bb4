// random_subblocks.hip — diagnostic: what does a wave pay for reading 4 random 16-vector sub-blocks of the byte copy
// (8 pieces of 256 contiguous bytes at 1 KB stride each, the select's exact rounds) from buffers of different sizes?
// hipcc --offload-arch=gfx950 -O3 random_subblocks.hip -o random_subblocks
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void __launch_bounds__(256) k(const uint4 *buf, uint64_t nblocks, int rounds, unsigned *out, int dependent) {
  const int lane = threadIdx.x & 63;
  const uint32_t wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  uint32_t state = wave * 2654435761u + 12345u, acc = 0;
  for (int r = 0; r < rounds; ++r) {
    state = state * 1664525u + 1013904223u;
    uint32_t s4 = state;
    // each group of 16 lanes: its own random block and sub-block
    s4 ^= (uint32_t)(lane >> 4) * 0x9E3779B9u;
    s4 = s4 * 1664525u + 1013904223u;
    const uint64_t b = (uint64_t)(s4 >> 8) % nblocks;
    const uint32_t sub = s4 & 3u;
    const uint4 *p = buf + b * 512 + 16 * sub + (lane & 15);  // block = 8 pieces x 64 vectors x 16 B
    uint4 x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = p[i * 64];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc += x[i].x ^ x[i].y ^ x[i].z ^ x[i].w;
    if (dependent) state ^= acc & 1u;  // (the next round's addresses wait for this round's data, as the select's do for its top-k)
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
int main() {
  unsigned *out;
  hipMalloc(&out, 2500 * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (uint64_t mb : {4ull, 128ull, 1024ull, 8192ull}) {
    uint4 *buf;
    const uint64_t bytes = mb << 20;
    if (hipMalloc(&buf, bytes) != hipSuccess) continue;
    hipMemset(buf, 1, bytes);
    const uint64_t nblocks = bytes / 8192;
    for (int dep = 0; dep < 2; ++dep)
      for (int waves_per : {2500}) {
        const int rounds = 16;
        for (int rep = 0; rep < 2; ++rep) {
          hipEventRecord(e0);
          hipLaunchKernelGGL(k, dim3(waves_per), dim3(256), 0, 0, buf, nblocks, rounds, out, dep);
          hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double nround = 2500.0 * 4 * rounds;
        printf("buffer %5llu MB, %s rounds: %.3f ms for %d rounds of 10000 waves = %.2f us per wave-round chip-wide (%.1f ns amortised), %.2f TB/s of useful bytes\n",
               (unsigned long long)mb, dep ? "dependent  " : "independent", ms, rounds, ms * 1e3 / rounds, ms * 1e6 / nround, nround * 8192 / (ms * 1e-3) / 1e12);
      }
    hipFree(buf);
  }
  return 0;
}
