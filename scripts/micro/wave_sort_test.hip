// wave_sort_test.hip — checks csrc/wave_sort.hpp on the GPU: every lane exchange against its definition, the sorting
// network against std::sort, FastTopK / FastTop128 against a host top-K under random offers (ties, infinities, NaN,
// negative values, empty lanes).   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I vector-indexer_amd/csrc ... && ./wave_sort_test
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

#include "wave_sort.hpp"

using namespace vi;

__global__ void exchange_kernel(uint32_t *out) {  // out[e][lane] = lane the value came from
  const int lane = threadIdx.x;
  const uint32_t x = (uint32_t)lane;
  out[0 * 64 + lane] = exchange_u32<Ex::X1>(x, lane);
  out[1 * 64 + lane] = exchange_u32<Ex::X2>(x, lane);
  out[2 * 64 + lane] = exchange_u32<Ex::X4>(x, lane);
  out[3 * 64 + lane] = exchange_u32<Ex::X8>(x, lane);
  out[4 * 64 + lane] = exchange_u32<Ex::X16>(x, lane);
  out[5 * 64 + lane] = exchange_u32<Ex::X32>(x, lane);
  out[6 * 64 + lane] = exchange_u32<Ex::M3>(x, lane);
  out[7 * 64 + lane] = exchange_u32<Ex::M7>(x, lane);
  out[8 * 64 + lane] = exchange_u32<Ex::M15>(x, lane);
  out[9 * 64 + lane] = exchange_u32<Ex::M31>(x, lane);
  out[10 * 64 + lane] = exchange_u32<Ex::M63>(x, lane);
}

__global__ void sort_kernel(uint64_t *keys, int rows) {
  const int lane = threadIdx.x & 63, w = blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
  if (w >= rows) return;
  uint64_t k = keys[(size_t)w * 64 + lane];
  wave_sort_u64(k, lane);
  keys[(size_t)w * 64 + lane] = k;
}

template <class Top>
__global__ void topk_kernel(const float *d, const uint32_t *p, int rounds, int K, float *od, uint32_t *op, int rows) {
  const int lane = threadIdx.x & 63, w = blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
  if (w >= rows) return;
  Top t;
  t.init();
  for (int r = 0; r < rounds; ++r) {
    const size_t i = ((size_t)w * rounds + r) * 64 + lane;
    t.offer_bulk(d[i], p[i], K);
  }
  for (int e = 0; e < Top::kEntries; ++e) {
    od[((size_t)w * 2 + e) * 64 + lane] = t.ent_d(e);
    op[((size_t)w * 2 + e) * 64 + lane] = t.ent_p(e);
  }
  if (lane == 0) od[((size_t)w * 2 + 1) * 64 + 63 + 0] = Top::kEntries == 1 ? t.kth(K) : od[((size_t)w * 2 + 1) * 64 + 63];
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 2; } } while (0)

__global__ void scan_kernel(const uint32_t *in, uint32_t *out) {  // one wave per 64 values
  const uint32_t i = blockIdx.x * 64 + threadIdx.x;
  out[i] = wave_incl_scan_u32(in[i]);
}

static uint32_t sortable_host(float x) {
  uint32_t b; memcpy(&b, &x, 4);
  if (x != x) return 0xFFFFFFFFu;
  return b ^ ((uint32_t)((int32_t)b >> 31) | 0x80000000u);
}

int main() {
  int bad = 0;
  {  // inclusive scan over the lanes
    const int rows = 512;
    std::vector<uint32_t> h(rows * 64), g(rows * 64);
    std::mt19937 r2(5);
    for (auto &v : h) v = r2() % 3 == 0 ? 0u : r2() % 1000u;
    for (int l = 0; l < 64; ++l) { h[l] = 1u; h[64 + l] = (uint32_t)l; }   // all ones; the lane index
    uint32_t *di, *dout; CK(hipMalloc(&di, h.size() * 4)); CK(hipMalloc(&dout, h.size() * 4));
    CK(hipMemcpy(di, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(scan_kernel, dim3(rows), dim3(64), 0, 0, di, dout);
    CK(hipMemcpy(g.data(), dout, g.size() * 4, hipMemcpyDeviceToHost));
    int b0 = 0;
    for (int w = 0; w < rows; ++w) {
      uint32_t run = 0;
      for (int l = 0; l < 64; ++l) {
        run += h[w * 64 + l];
        if (g[w * 64 + l] != run) { if (b0 < 5) printf("scan row %d lane %d: got %u want %u\n", w, l, g[w * 64 + l], run); ++b0; }
      }
    }
    printf("scan: %d wrong prefix sums\n", b0);
    bad += b0;
    CK(hipFree(di)); CK(hipFree(dout));
  }
  {  // exchanges
    uint32_t *d; CK(hipMalloc(&d, 11 * 64 * 4));
    hipLaunchKernelGGL(exchange_kernel, dim3(1), dim3(64), 0, 0, d);
    std::vector<uint32_t> h(11 * 64); CK(hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost));
    const int x[11] = {1, 2, 4, 8, 16, 32, 3, 7, 15, 31, 63};
    for (int e = 0; e < 11; ++e)
      for (int l = 0; l < 64; ++l)
        if (h[e * 64 + l] != (uint32_t)(l ^ x[e])) { if (bad < 10) printf("exchange %d lane %d: got %u want %d\n", x[e], l, h[e * 64 + l], l ^ x[e]); ++bad; }
    printf("exchanges: %s\n", bad ? "FAILED" : "ok");
    CK(hipFree(d));
  }
  std::mt19937_64 rng(7);
  {  // sort
    const int rows = 4096;
    std::vector<uint64_t> h(rows * 64), ref;
    for (auto &v : h) { v = rng(); if (rng() % 5 == 0) v &= 0xFFull; if (rng() % 7 == 0) v = h[0]; }
    ref = h;
    uint64_t *d; CK(hipMalloc(&d, h.size() * 8)); CK(hipMemcpy(d, h.data(), h.size() * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(sort_kernel, dim3(rows / 4), dim3(256), 0, 0, d, rows);
    CK(hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost));
    int b2 = 0;
    for (int r = 0; r < rows; ++r) {
      std::sort(ref.begin() + r * 64, ref.begin() + r * 64 + 64);
      if (memcmp(&ref[r * 64], &h[r * 64], 512)) ++b2;
    }
    printf("sort: %d of %d rows wrong\n", b2, rows);
    bad += b2;
    CK(hipFree(d));
  }
  for (int variant = 0; variant < 2; ++variant) {  // top-K
    const int rows = 2048, rounds = 9;
    std::vector<float> d(rows * rounds * 64);
    std::vector<uint32_t> p(d.size());
    std::uniform_real_distribution<float> U(-100.f, 100.f);
    for (int w = 0; w < rows; ++w)
      for (int i = 0; i < rounds * 64; ++i) {
        const size_t o = (size_t)w * rounds * 64 + i;
        float v = U(rng);
        const int m = (int)(rng() % 16);
        if (m == 0) v = std::floor(v);                 // ties
        if (m == 1) v = INFINITY;
        if (m == 2) v = NAN;
        if (m == 3) v = -NAN;
        if (m == 4) v = 0.0f;
        if (w % 3 == 0 && m < 12) v = std::floor(v / 20.f);  // masses of ties
        p[o] = (uint32_t)i * 7u + (uint32_t)w;          // unique per row
        d[o] = v;
        if (m == 5 || (w % 5 == 0 && i > 70)) { d[o] = INFINITY; p[o] = 0xFFFFFFFFu; }  // empty lane
      }
    float *dd, *od; uint32_t *dp, *op;
    CK(hipMalloc(&dd, d.size() * 4)); CK(hipMalloc(&dp, p.size() * 4)); CK(hipMalloc(&od, rows * 128 * 4)); CK(hipMalloc(&op, rows * 128 * 4));
    CK(hipMemcpy(dd, d.data(), d.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dp, p.data(), p.size() * 4, hipMemcpyHostToDevice));
    for (int K : {1, 10, 33, 64, 100, 128}) {
      if (variant == 0 && K > 64) continue;
      if (variant == 0) hipLaunchKernelGGL(topk_kernel<FastTopK>, dim3(rows / 4), dim3(256), 0, 0, dd, dp, rounds, K, od, op, rows);
      else hipLaunchKernelGGL(topk_kernel<FastTop128>, dim3(rows / 4), dim3(256), 0, 0, dd, dp, rounds, K, od, op, rows);
      std::vector<float> hd(rows * 128); std::vector<uint32_t> hp(rows * 128);
      CK(hipMemcpy(hd.data(), od, hd.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hp.data(), op, hp.size() * 4, hipMemcpyDeviceToHost));
      int b3 = 0;
      for (int w = 0; w < rows; ++w) {
        std::vector<std::pair<uint64_t, float>> c;
        for (int i = 0; i < rounds * 64; ++i) {
          const size_t o = (size_t)w * rounds * 64 + i;
          if (d[o] != d[o]) continue;                                   // NaN never enters
          if (p[o] == 0xFFFFFFFFu && d[o] == INFINITY) continue;        // empty lane
          c.push_back({((uint64_t)sortable_host(d[o]) << 32) | p[o], d[o]});
        }
        std::sort(c.begin(), c.end());
        for (int i = 0; i < K; ++i) {
          const float gd = hd[(size_t)w * 128 + i];
          const uint32_t gp = hp[(size_t)w * 128 + i];
          const bool have = i < (int)c.size();
          const float wd = have ? c[i].second : INFINITY;
          const uint32_t wp = have ? (uint32_t)c[i].first : 0xFFFFFFFFu;
          if (gp != wp || memcmp(&gd, &wd, 4)) { if (b3 < 5) printf("top%d row %d entry %d: got (%g,%u) want (%g,%u)\n", K, w, i, gd, gp, wd, wp); ++b3; }
        }
      }
      printf("%s K=%d: %d wrong entries\n", variant ? "FastTop128" : "FastTopK", K, b3);
      bad += b3;
    }
    CK(hipFree(dd)); CK(hipFree(dp)); CK(hipFree(od)); CK(hipFree(op));
  }
  printf(bad ? "FAILED\n" : "ALL OK\n");
  return bad ? 1 : 0;
}
