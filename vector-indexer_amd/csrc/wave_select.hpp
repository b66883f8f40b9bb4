// wave_select.hpp — wave64 primitives and the wave-resident sorted top-K used by the search kernels.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "device_index.hpp"

namespace vi {

// ------------------------------------------------------------------------------------------
// wave primitives
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float readlane_f(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
__device__ __forceinline__ uint32_t readlane_u(uint32_t v, int lane) {
  return (uint32_t)__builtin_amdgcn_readlane((int)v, lane);
}
// value of lane-1 (DPP wave_shr:1); lane 0 receives `fill`
__device__ __forceinline__ float shr1_f(float v, float fill) {
  return __int_as_float(
      __builtin_amdgcn_update_dpp(__float_as_int(fill), __float_as_int(v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ uint32_t shr1_u(uint32_t v, uint32_t fill) {
  return (uint32_t)__builtin_amdgcn_update_dpp((int)fill, (int)v, 0x138, 0xf, 0xf, false);
}
__device__ __forceinline__ uint64_t wave_min_u64(uint64_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, o);
    const uint32_t hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), o);
    const uint64_t t = ((uint64_t)hi << 32) | lo;
    v = t < v ? t : v;
  }
  return v;
}

// ------------------------------------------------------------------------------------------
// Wave-resident sorted top-K (K <= 64): lane i holds the i-th best (dist, pos) pair in
// ascending (dist, pos) order.  `thr`/`thrp` cache entry K-1 (wave-uniform).
// ------------------------------------------------------------------------------------------
struct WaveTopK {
  float d;
  uint32_t p;
  float thr;
  uint32_t thrp;
  __device__ __forceinline__ void init() {
    d = INFINITY; p = kNoPos; thr = INFINITY; thrp = kNoPos;
  }
  // Offer one candidate per lane (dist, pos); pos == kNoPos marks an invalid lane.
  // Candidates beat entry K-1 iff (dist,pos) < (thr,thrp) lexicographically.
  __device__ __forceinline__ void offer(float dist, uint32_t pos, int K) {
    bool pass = (dist < thr) || (dist == thr && pos < thrp);
    uint64_t mask = __ballot(pass);
    while (mask) {
      const int src = __builtin_ctzll(mask);
      const float cd = readlane_f(dist, src);
      const uint32_t cp = readlane_u(pos, src);
      // entries greater than the candidate shift one lane to the right
      const bool gt = (d > cd) || (d == cd && p > cp);
      const float ud = shr1_f(d, -INFINITY);
      const uint32_t up = shr1_u(p, 0u);
      const bool ugt = (ud > cd) || (ud == cd && up > cp);
      d = gt ? (ugt ? ud : cd) : d;
      p = gt ? (ugt ? up : cp) : p;
      thr = readlane_f(d, K - 1);
      thrp = readlane_u(p, K - 1);
      pass = (dist < thr) || (dist == thr && pos < thrp);
      const uint64_t rest = (src == 63) ? 0ull : (~0ull << (src + 1));
      mask = __ballot(pass) & rest;
    }
  }
  // ---- bulk variant (select kernels): when more than a few lanes beat entry K-1, one bitonic sort + merge
  //      (straight-line cross-lane code) replaces that many serial insertions ----
  static __device__ __forceinline__ void cmpx(float &v, uint32_t &k, int j, bool want_min) {
    const float pv = __shfl_xor(v, j);
    const uint32_t pk = (uint32_t)__shfl_xor((int)k, j);
    const bool p_less = (pv < v) || (pv == v && pk < k);
    const bool take = want_min == p_less;
    v = take ? pv : v;
    k = take ? pk : k;
  }
  __device__ __forceinline__ void offer_bulk(float dist, uint32_t pos, int K) {
    const bool pass = (dist < thr) || (dist == thr && pos < thrp);
    const uint64_t mask = __ballot(pass);
    if (!mask) return;
    if (__popcll(mask) <= 3) { offer(dist, pos, K); return; }
    const int lane = (int)(threadIdx.x & 63u);
    float v = pass ? dist : INFINITY;
    uint32_t k = pass ? pos : kNoPos;
    // bitonic sort of the 64 incoming pairs, DESCENDING
#pragma unroll
    for (int kk = 2; kk <= 64; kk <<= 1) {
#pragma unroll
      for (int j = kk >> 1; j > 0; j >>= 1) {
        const bool lower = (lane & j) == 0;
        const bool desc_block = (lane & kk) == 0;  // kk == 64: every lane
        cmpx(v, k, j, lower != desc_block);
      }
    }
    // ascending list (d, p) vs descending incoming: the pairwise minima are the 64 smallest of the union (bitonic)
    {
      const bool in_less = (v < d) || (v == d && k < p);
      d = in_less ? v : d;
      p = in_less ? k : p;
    }
#pragma unroll
    for (int j = 32; j > 0; j >>= 1) cmpx(d, p, j, (lane & j) == 0);
    thr = readlane_f(d, K - 1);
    thrp = readlane_u(p, K - 1);
  }
  // ---- the interface the select kernels are written against (WaveTop128 below offers the same) ----
  static constexpr int kEntries = 1;  // entries per lane
  __device__ __forceinline__ float kth(int K) const { return readlane_f(d, K - 1); }
  __device__ __forceinline__ float ent_d(int) const { return d; }       // entry e of this lane = rank 64*e + lane
  __device__ __forceinline__ uint32_t ent_p(int) const { return p; }
  // keep only the entries flagged keep_e (of the first K): the survivors close ranks
  __device__ __forceinline__ void rebuild(bool keep0, bool, int K) {
    const float kd = keep0 ? d : INFINITY;
    const uint32_t kp = keep0 ? p : kNoPos;
    init();
    offer_bulk(kd, kp, K);
  }
};

// The same for K <= 128: lane i holds entries i (d0, p0) and 64 + i (d1, p1) of the ascending list.  Only the bulk
// offer exists (the select kernels use nothing else): the 64 incoming pairs are sorted descending; their pairwise
// minima with the upper half are the 64 smallest of (upper half, incoming) — a bitonic sequence, merged ascending;
// that half, reversed, against the lower half gives the new lower half (minima) and upper half (maxima), both
// bitonic again: 39 compare-exchange steps against 27 of the 64-entry form.
struct WaveTop128 {
  float d0, d1;
  uint32_t p0, p1;
  float thr;
  uint32_t thrp;
  static constexpr int kEntries = 2;
  __device__ __forceinline__ void init() {
    d0 = d1 = INFINITY; p0 = p1 = kNoPos; thr = INFINITY; thrp = kNoPos;
  }
  __device__ __forceinline__ float kth(int K) const { return K <= 64 ? readlane_f(d0, K - 1) : readlane_f(d1, K - 65); }
  __device__ __forceinline__ float ent_d(int e) const { return e ? d1 : d0; }
  __device__ __forceinline__ uint32_t ent_p(int e) const { return e ? p1 : p0; }
  __device__ __forceinline__ void set_thr(int K) {
    thr = K <= 64 ? readlane_f(d0, K - 1) : readlane_f(d1, K - 65);
    thrp = K <= 64 ? readlane_u(p0, K - 1) : readlane_u(p1, K - 65);
  }
  __device__ __forceinline__ void offer_bulk(float dist, uint32_t pos, int K) {
    const bool pass = (dist < thr) || (dist == thr && pos < thrp);
    if (!__ballot(pass)) return;
    const int lane = (int)(threadIdx.x & 63u);
    float v = pass ? dist : INFINITY;
    uint32_t k = pass ? pos : kNoPos;
#pragma unroll
    for (int kk = 2; kk <= 64; kk <<= 1) {  // bitonic sort of the incoming pairs, DESCENDING
#pragma unroll
      for (int j = kk >> 1; j > 0; j >>= 1) {
        const bool lower = (lane & j) == 0;
        const bool desc_block = (lane & kk) == 0;
        WaveTopK::cmpx(v, k, j, lower != desc_block);
      }
    }
    {  // upper half (ascending) vs incoming (descending): the minima are the 64 smallest of the two
      const bool in_less = (v < d1) || (v == d1 && k < p1);
      d1 = in_less ? v : d1;
      p1 = in_less ? k : p1;
    }
#pragma unroll
    for (int j = 32; j > 0; j >>= 1) WaveTopK::cmpx(d1, p1, j, (lane & j) == 0);
    {  // lower half (ascending) vs that half reversed (descending): minima stay below, maxima go above
      const float rv = __shfl(d1, 63 - lane);
      const uint32_t rk = (uint32_t)__shfl((int)p1, 63 - lane);
      const bool r_less = (rv < d0) || (rv == d0 && rk < p0);
      d1 = r_less ? d0 : rv;
      p1 = r_less ? p0 : rk;
      d0 = r_less ? rv : d0;
      p0 = r_less ? rk : p0;
    }
#pragma unroll
    for (int j = 32; j > 0; j >>= 1) {
      WaveTopK::cmpx(d0, p0, j, (lane & j) == 0);
      WaveTopK::cmpx(d1, p1, j, (lane & j) == 0);
    }
    set_thr(K);
  }
  __device__ __forceinline__ void rebuild(bool keep0, bool keep1, int K) {
    const float a = keep0 ? d0 : INFINITY, b = keep1 ? d1 : INFINITY;
    const uint32_t ap = keep0 ? p0 : kNoPos, bp = keep1 ? p1 : kNoPos;
    init();
    offer_bulk(a, ap, K);
    offer_bulk(b, bp, K);
  }
};


// Reference candidate order of a query's probes (src/ivf_index.rs:223-262): shards are visited in order of
// first appearance in the probe list, probes of one shard in rank order.  Lane i < found holds probe i
// (list id `mylist`); returns the rank g of probe i under the key (first_appearance(shard_i), i).
__device__ __forceinline__ uint32_t probe_candidate_order(int lane, uint32_t found, uint32_t mylist,
                                                          const uint32_t *list_shard) {
  const bool live = (uint32_t)lane < found;
  const uint32_t shard = live ? list_shard[mylist] : kNoPos;
  // one step per DISTINCT shard, in order of first appearance (the lowest lane still waiting leads): its probes take the
  // next ranks in lane order.  (Two passes over all probes with a readlane each were an eighth of the coarse select.)
  const uint64_t below = (1ull << lane) - 1ull;
  uint64_t todo = __ballot(live);
  uint32_t g = found, base = 0;  // (lanes without a probe: `found`, as the rank under the key kNoPos)
  while (todo) {
    const uint32_t s = readlane_u(shard, __builtin_ctzll(todo));
    const bool mine = live && shard == s;
    const uint64_t same = __ballot(mine);
    if (mine) g = base + (uint32_t)__popcll(same & below);
    base += (uint32_t)__popcll(same);
    todo &= ~same;
  }
  return g;
}

}  // namespace vi
