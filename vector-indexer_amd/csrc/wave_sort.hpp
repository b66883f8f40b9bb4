// wave_sort.hpp — wave64 sorting network on packed 64-bit keys whose lane exchanges are DPP moves and gfx950's
// v_permlane{16,32}_swap instead of ds_bpermute round trips through the LDS crossbar.
//
// The select kernels keep the K best (value, tie key) pairs of a query sorted across the lanes of a wave and merge 64
// offered pairs at a time: a bitonic sort of the offer (21 compare-exchange steps) + a 6-step merge.  With __shfl_xor
// every step was two ds_bpermute_b32 (≈ 100 cycles of latency each way) and ≈ 30 instructions of two-word comparison:
// ≈ 800 instructions and ≈ 6 000 cycles of dependent latency per offer, 15-20 offers per query.  Here a pair is ONE
// u64 key — (order-preserving image of the f32 value) << 32 | tie key — so a step is: partner = two DPP moves,
// v_cmp_lt_u64, one scalar xnor with the step's lane mask, two v_cndmask.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "device_index.hpp"

namespace vi {

// order-preserving map of f32 onto u32: a < b (as floats) => key(a) < key(b); NaN of either sign maps above +inf
__device__ __forceinline__ uint32_t f32_sortable(float x) {
  const uint32_t b = __float_as_uint(x);
  const uint32_t s = b ^ ((uint32_t)((int32_t)b >> 31) | 0x80000000u);
  return x != x ? 0xFFFFFFFFu : s;
}
__device__ __forceinline__ float sortable_f32(uint32_t s) {
  return __uint_as_float((s & 0x80000000u) ? (s ^ 0x80000000u) : ~s);
}
__device__ __forceinline__ uint64_t pack_key(float v, uint32_t tie) { return ((uint64_t)f32_sortable(v) << 32) | tie; }

// ---- lane exchanges --------------------------------------------------------------------------------------------------
enum class Ex { X1, X2, X4, X8, X16, X32, M3, M7, M15, M31, M63 };  // Xj: lane ^ j;  Mj: lane ^ j with j = 2^n - 1 (mirror)

template <int CTRL, int BANK = 0xf>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t old, uint32_t x) {
  return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)x, CTRL, 0xf, BANK, false);
}

template <Ex E>
__device__ __forceinline__ uint32_t exchange_u32(uint32_t x, int lane) {
  if constexpr (E == Ex::X1) return dpp_u32<0xB1>(x, x);        // quad_perm [1,0,3,2]
  else if constexpr (E == Ex::X2) return dpp_u32<0x4E>(x, x);   // quad_perm [2,3,0,1]
  else if constexpr (E == Ex::M3) return dpp_u32<0x1B>(x, x);   // quad_perm [3,2,1,0]
  else if constexpr (E == Ex::X4) {                             // banks 0,2 (lanes 0-3, 8-11 of a row) take lane+4, banks 1,3 lane-4
    const uint32_t t = dpp_u32<0x12C, 0x5>(x, x);               // row_ror:12 = lane + 4 (mod 16)
    return dpp_u32<0x124, 0xA>(t, x);                           // row_ror:4  = lane - 4 (mod 16)
  } else if constexpr (E == Ex::X8) return dpp_u32<0x128>(x, x);  // row_ror:8
  else if constexpr (E == Ex::M7) return dpp_u32<0x141>(x, x);    // row_half_mirror
  else if constexpr (E == Ex::M15) return dpp_u32<0x140>(x, x);   // row_mirror
  else if constexpr (E == Ex::X16) {
    // v_permlane16_swap: rows 1, 3 of the first operand <-> rows 0, 2 of the second
    const auto r = __builtin_amdgcn_permlane16_swap(x, x, false, false);
    return (lane & 16) ? r[0] : r[1];
  } else if constexpr (E == Ex::X32) {
    // v_permlane32_swap: upper half of the first operand <-> lower half of the second
    const auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false);
    return (lane & 32) ? r[0] : r[1];
  } else if constexpr (E == Ex::M31) return exchange_u32<Ex::M15>(exchange_u32<Ex::X16>(x, lane), lane);
  else return exchange_u32<Ex::M15>(exchange_u32<Ex::X16>(exchange_u32<Ex::X32>(x, lane), lane), lane);  // M63
}

template <Ex E>
__device__ __forceinline__ uint64_t exchange_u64(uint64_t k, int lane) {
  const uint32_t lo = exchange_u32<E>((uint32_t)k, lane), hi = exchange_u32<E>((uint32_t)(k >> 32), lane);
  return ((uint64_t)hi << 32) | lo;
}

// inclusive prefix sum over the 64 lanes: shifts within the rows of 16 lanes, then the row totals (row_bcast) — six DPP
// adds, no LDS crossbar (a __shfl_up loop is six dependent ds_bpermute round trips)
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t x) {
  x += dpp_u32<0x111>(0u, x);  // row_shr:1  (lanes without a source take the `old` operand: 0)
  x += dpp_u32<0x112>(0u, x);  // row_shr:2
  x += dpp_u32<0x114>(0u, x);  // row_shr:4
  x += dpp_u32<0x118>(0u, x);  // row_shr:8
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xA, 0xF, false);  // row_bcast:15 into rows 1, 3
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xC, 0xF, false);  // row_bcast:31 into rows 2, 3
  return x;
}

// compare-exchange with the partner lane: the lane whose `keep_min` is set keeps the smaller key
template <Ex E>
__device__ __forceinline__ void cmpx_u64(uint64_t &k, int lane, bool keep_min) {
  const uint64_t pk = exchange_u64<E>(k, lane);
  k = ((pk < k) == keep_min) ? pk : k;
}

// the 64 keys of a wave in ascending lane order
__device__ __forceinline__ void wave_sort_u64(uint64_t &k, int lane) {
  cmpx_u64<Ex::X1>(k, lane, !(lane & 1));
  cmpx_u64<Ex::M3>(k, lane, !(lane & 2));
  cmpx_u64<Ex::X1>(k, lane, !(lane & 1));
  cmpx_u64<Ex::M7>(k, lane, !(lane & 4));
  cmpx_u64<Ex::X2>(k, lane, !(lane & 2));
  cmpx_u64<Ex::X1>(k, lane, !(lane & 1));
  cmpx_u64<Ex::M15>(k, lane, !(lane & 8));
  cmpx_u64<Ex::X4>(k, lane, !(lane & 4));
  cmpx_u64<Ex::X2>(k, lane, !(lane & 2));
  cmpx_u64<Ex::X1>(k, lane, !(lane & 1));
  cmpx_u64<Ex::M31>(k, lane, !(lane & 16));
  cmpx_u64<Ex::X8>(k, lane, !(lane & 8));
  cmpx_u64<Ex::X4>(k, lane, !(lane & 4));
  cmpx_u64<Ex::X2>(k, lane, !(lane & 2));
  cmpx_u64<Ex::X1>(k, lane, !(lane & 1));
  cmpx_u64<Ex::M63>(k, lane, !(lane & 32));
  cmpx_u64<Ex::X16>(k, lane, !(lane & 16));
  cmpx_u64<Ex::X8>(k, lane, !(lane & 8));
  cmpx_u64<Ex::X4>(k, lane, !(lane & 4));
  cmpx_u64<Ex::X2>(k, lane, !(lane & 2));
  cmpx_u64<Ex::X1>(k, lane, !(lane & 1));
}
// a bitonic sequence (ascending then descending, or any rotation of one) into ascending order
__device__ __forceinline__ void wave_merge_u64(uint64_t &k, int lane) {
  cmpx_u64<Ex::X32>(k, lane, !(lane & 32));
  cmpx_u64<Ex::X16>(k, lane, !(lane & 16));
  cmpx_u64<Ex::X8>(k, lane, !(lane & 8));
  cmpx_u64<Ex::X4>(k, lane, !(lane & 4));
  cmpx_u64<Ex::X2>(k, lane, !(lane & 2));
  cmpx_u64<Ex::X1>(k, lane, !(lane & 1));
}
__device__ __forceinline__ uint64_t readlane_u64(uint64_t k, int src) {
  return ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(k >> 32), src) << 32) |
         (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)k, src);
}

// ------------------------------------------------------------------------------------------
// Wave-resident sorted top-K (K <= 64) on packed keys: lane i holds the i-th smallest key.  Same interface as
// WaveTopK / WaveTop128 of wave_select.hpp (the select kernels are templates over it).
// ------------------------------------------------------------------------------------------
struct FastTopK {
  uint64_t key, thr;  // thr = key of entry K-1 (wave-uniform)
  static constexpr int kEntries = 1;
  static constexpr int kSeqInsert = 8;  // newcomers inserted one by one up to here
  static constexpr uint64_t kEmpty = ((uint64_t)0xFF800000u << 32) | 0xFFFFFFFFu;  // (+inf, kNoPos)
  __device__ __forceinline__ void init() { key = kEmpty; thr = kEmpty; }
  __device__ __forceinline__ float kth(int K) const { return sortable_f32((uint32_t)(readlane_u64(key, K - 1) >> 32)); }
  __device__ __forceinline__ float ent_d(int) const { return sortable_f32((uint32_t)(key >> 32)); }
  __device__ __forceinline__ uint32_t ent_p(int) const { return (uint32_t)key; }
  // one candidate per lane; pos == kNoPos with dist == +inf marks an empty lane (it never beats entry K-1)
  __device__ __forceinline__ void offer_bulk(float dist, uint32_t pos, int K) {
    const int lane = (int)(threadIdx.x & 63u);
    uint64_t v = pack_key(dist, pos);
    const bool pass = v < thr;
    const uint64_t mask = __ballot(pass);
    if (!mask) return;
    v = pass ? v : kEmpty;
    if (__popcll(mask) <= kSeqInsert) {  // a few newcomers, one after the other: entries above each move up a lane
      // (~10 instructions each against ~300 for the sort and merge below; once the list has filled, a round of 64
      //  offers brings K * 64 / (offered so far) newcomers on average — a handful from the second round on)
      for (uint64_t m = mask; m; m &= m - 1ull) {
        const uint64_t c = readlane_u64(v, __builtin_ctzll(m));
        // wave_shr:1 — lane 0 has no source and keeps the `old` operand, 0: below every key
        const uint32_t ulo = dpp_u32<0x138>(0u, (uint32_t)key), uhi = dpp_u32<0x138>(0u, (uint32_t)(key >> 32));
        const uint64_t up = ((uint64_t)uhi << 32) | ulo;
        key = key > c ? (up > c ? up : c) : key;
      }
    } else {
      wave_sort_u64(v, lane);
      if (readlane_u64(key, 0) == kEmpty) {  // an empty list (every first offer): the sorted offer is the list
        key = v;
      } else {
        const uint64_t rv = exchange_u64<Ex::M63>(v, lane);  // descending: min(list, reversed offer) holds the 64 smallest
        key = rv < key ? rv : key;
        wave_merge_u64(key, lane);
      }
    }
    thr = readlane_u64(key, K - 1);
  }
  // keep only the entries flagged keep (of the first K): the survivors close ranks
  __device__ __forceinline__ void rebuild(bool keep0, bool, int K) {
    const int lane = (int)(threadIdx.x & 63u);
    key = keep0 ? key : kEmpty;
    wave_sort_u64(key, lane);
    thr = readlane_u64(key, K - 1);
  }
};

// K <= 128: lane i holds entries i (k0) and 64 + i (k1) of the ascending list
struct FastTop128 {
  uint64_t k0, k1, thr;
  static constexpr int kEntries = 2;
  __device__ __forceinline__ void init() { k0 = k1 = thr = FastTopK::kEmpty; }
  __device__ __forceinline__ uint64_t entry_key(int K) const { return K <= 64 ? readlane_u64(k0, K - 1) : readlane_u64(k1, K - 65); }
  __device__ __forceinline__ float kth(int K) const { return sortable_f32((uint32_t)(entry_key(K) >> 32)); }
  __device__ __forceinline__ float ent_d(int e) const { return sortable_f32((uint32_t)((e ? k1 : k0) >> 32)); }
  __device__ __forceinline__ uint32_t ent_p(int e) const { return (uint32_t)(e ? k1 : k0); }
  __device__ __forceinline__ void offer_bulk(float dist, uint32_t pos, int K) {
    const int lane = (int)(threadIdx.x & 63u);
    uint64_t v = pack_key(dist, pos);
    const bool pass = v < thr;
    if (!__ballot(pass)) return;
    v = pass ? v : FastTopK::kEmpty;
    wave_sort_u64(v, lane);
    {  // upper half (ascending) vs the offer reversed (descending): the minima are the 64 smallest of the two, bitonic
      const uint64_t rv = exchange_u64<Ex::M63>(v, lane);
      k1 = rv < k1 ? rv : k1;
      wave_merge_u64(k1, lane);
    }
    {  // lower half (ascending) vs that half reversed: minima stay below, maxima go above; both bitonic again
      const uint64_t rv = exchange_u64<Ex::M63>(k1, lane);
      const bool less = rv < k0;
      k1 = less ? k0 : rv;
      k0 = less ? rv : k0;
      wave_merge_u64(k0, lane);
      wave_merge_u64(k1, lane);
    }
    thr = entry_key(K);
  }
  __device__ __forceinline__ void rebuild(bool keep0, bool keep1, int K) {
    const uint64_t a = keep0 ? k0 : FastTopK::kEmpty, b = keep1 ? k1 : FastTopK::kEmpty;
    init();
    offer_bulk(sortable_f32((uint32_t)(a >> 32)), (uint32_t)a, K);
    offer_bulk(sortable_f32((uint32_t)(b >> 32)), (uint32_t)b, K);
  }
};

}  // namespace vi
