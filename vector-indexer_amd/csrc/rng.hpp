// rng.hpp — host RNG of the k-means control loop.
//
// The reference draws every random decision of k-means from rand 0.8.5's
// StdRng::seed_from_u64 (src/kmeans.rs:31,80,170,240,591).  rand is a Cargo
// dependency that is not vendored under the reference tree, so this is a
// restatement of the crates' published algorithms (rand 0.8.5, rand_chacha
// 0.3.1, rand_core 0.6.4): ChaCha12 block function, PCG32 seed expansion,
// widening-multiply range sampling, Fisher-Yates shuffle, reservoir
// choose_multiple and WeightedIndex<f32>.  All of it stays on the host: the GPU
// path and the CPU oracle consume the same stream and differ only in where the
// distances are computed.
#pragma once
#include <cstdint>
#include <cstring>
#include <vector>

namespace vi {

class StdRng {
 public:
  explicit StdRng(uint64_t seed) { seed_from_u64(seed); }

  uint32_t next_u32() {
    if (index_ >= 64) { refill(); index_ = 0; }
    return buf_[index_++];
  }

  uint64_t next_u64() {
    const uint32_t i = index_;
    if (i < 63) {
      index_ += 2;
      return ((uint64_t)buf_[i + 1] << 32) | buf_[i];
    }
    if (i >= 64) {
      refill();
      index_ = 2;
      return ((uint64_t)buf_[1] << 32) | buf_[0];
    }
    const uint64_t lo = buf_[63];
    refill();
    index_ = 1;
    return ((uint64_t)buf_[0] << 32) | lo;
  }

  // rng.gen_range(low..high) for usize
  uint64_t gen_range(uint64_t low, uint64_t high) {
    const uint64_t range = high - low;
    if (range == 0) return next_u64();
    const uint64_t zone = (range << __builtin_clzll(range)) - 1;
    for (;;) {
      const unsigned __int128 m = (unsigned __int128)next_u64() * range;
      if ((uint64_t)m <= zone) return low + (uint64_t)(m >> 64);
    }
  }

  // rand::seq::gen_index
  uint64_t gen_index(uint64_t ubound) {
    if (ubound <= 0xFFFFFFFFull) {
      const uint32_t range = (uint32_t)ubound;
      const uint32_t zone = (range << __builtin_clz(range)) - 1;
      for (;;) {
        const uint64_t m = (uint64_t)next_u32() * range;
        if ((uint32_t)m <= zone) return m >> 32;
      }
    }
    return gen_range(0, ubound);
  }

  template <typename T>
  void shuffle(T *v, uint64_t n) {
    for (uint64_t i = n; i-- > 1;) {
      const uint64_t j = gen_index(i + 1);
      T t = v[i]; v[i] = v[j]; v[j] = t;
    }
  }

  // (0..n).choose_multiple(rng, amount)
  std::vector<uint64_t> choose_multiple_range(uint64_t n, uint64_t amount) {
    std::vector<uint64_t> res;
    const uint64_t len = amount < n ? amount : n;
    res.reserve(len);
    for (uint64_t i = 0; i < len; ++i) res.push_back(i);
    if (len == amount)
      for (uint64_t i = 0; amount + i < n; ++i) {
        const uint64_t k = gen_index(i + 1 + amount);
        if (k < amount) res[k] = amount + i;
      }
    return res;
  }

  // WeightedIndex::<f32>::new(w).sample(rng); cum is scratch of n floats
  uint64_t weighted_index(const float *w, uint64_t n, float *cum) {
    float total = w[0];
    for (uint64_t i = 1; i < n; ++i) { cum[i - 1] = total; total += w[i]; }
    const float max_rand = bits_to_f32((127u << 23) | 0x7FFFFFu) - 1.0f;
    float scale = total;  // high - low with low = 0
    while (!(scale * max_rand + 0.0f < total)) scale = bits_to_f32(f32_to_bits(scale) - 1);
    const float v01 = bits_to_f32((127u << 23) | (next_u32() >> 9)) - 1.0f;
    const float chosen = v01 * scale + 0.0f;
    uint64_t lo = 0, hi = n - 1;
    while (lo < hi) {
      const uint64_t mid = lo + (hi - lo) / 2;
      if (cum[mid] <= chosen) lo = mid + 1; else hi = mid;
    }
    return lo;
  }

 private:
  static float bits_to_f32(uint32_t b) { float f; std::memcpy(&f, &b, 4); return f; }
  static uint32_t f32_to_bits(float f) { uint32_t b; std::memcpy(&b, &f, 4); return b; }
  static uint32_t rotl(uint32_t x, int n) { return (x << n) | (x >> (32 - n)); }

  void seed_from_u64(uint64_t state) {
    for (int i = 0; i < 8; ++i) {
      state = state * 6364136223846793005ULL + 11634580027462260723ULL;
      const uint32_t xs = (uint32_t)(((state >> 18) ^ state) >> 27);
      const uint32_t rot = (uint32_t)(state >> 59);
      key_[i] = (xs >> rot) | (xs << ((32 - rot) & 31));
    }
    counter_ = 0;
    index_ = 64;
  }

  void block(uint64_t ctr, uint32_t *out) const {
    uint32_t s[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u};
    for (int i = 0; i < 8; ++i) s[4 + i] = key_[i];
    s[12] = (uint32_t)ctr; s[13] = (uint32_t)(ctr >> 32); s[14] = 0; s[15] = 0;
    uint32_t x[16];
    std::memcpy(x, s, sizeof(s));
    auto qr = [&](int a, int b, int c, int d) {
      x[a] += x[b]; x[d] = rotl(x[d] ^ x[a], 16);
      x[c] += x[d]; x[b] = rotl(x[b] ^ x[c], 12);
      x[a] += x[b]; x[d] = rotl(x[d] ^ x[a], 8);
      x[c] += x[d]; x[b] = rotl(x[b] ^ x[c], 7);
    };
    for (int r = 0; r < 6; ++r) {  // 12 rounds
      qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15);
      qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14);
    }
    for (int i = 0; i < 16; ++i) out[i] = x[i] + s[i];
  }

  void refill() {
    for (int b = 0; b < 4; ++b) block(counter_ + b, buf_ + 16 * b);
    counter_ += 4;
  }

  uint32_t key_[8];
  uint64_t counter_;
  uint32_t buf_[64];
  uint32_t index_;
};

}  // namespace vi
