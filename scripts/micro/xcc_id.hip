// xcc_id.hip — diagnostic: which XCD does workgroup b of a 1-D grid run on?  (HW_REG_XCC_ID, gfx942 / gfx950)
// hipcc --offload-arch=gfx950 -O3 xcc_id.hip -o xcc_id
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(unsigned *out) {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  if (threadIdx.x == 0) out[blockIdx.x] = v;
}
int main() {
  const int n = 2048;
  unsigned *d;
  hipMalloc(&d, n * 4);
  for (int threads : {64, 256}) {
    hipLaunchKernelGGL(k, dim3(n), dim3(threads), 0, 0, d);
    std::vector<unsigned> h(n);
    hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost);
    int agree = 0;
    for (int b = 0; b < n; ++b) agree += ((h[b] & 0xF) == (unsigned)(b % 8));
    printf("block %d threads: XCC_ID & 15 == blockIdx %% 8 for %d of %d workgroups; first 16:", threads, agree, n);
    for (int b = 0; b < 16; ++b) printf(" %u", h[b] & 0xF);
    printf("\n");
  }
  return 0;
}
