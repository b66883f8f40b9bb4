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

// bulk ChaCha12 keystream of a key: 16-word blocks [first_block, first_block + nblocks), valid until the next call;
// nullptr = failure.  kmeans.hip produces it on the GPU; HostKeystream below is the plain host version.
struct KeystreamSource {
  virtual ~KeystreamSource() = default;
  virtual const uint32_t *blocks(const uint32_t key[8], uint64_t first_block, uint64_t nblocks) = 0;
};

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

  // The first `take` entries of what shuffle() makes of the identity permutation 0..n-1 — all sample_batch
  // (src/kmeans.rs:722-726: shuffle all of 0..n, keep the first batch_size) ever reads — consuming exactly the stream
  // shuffle() consumes, without permuting a 4n-byte array at random:
  //   1. the keystream comes in bulk from `ks` (ChaCha is counter mode: the GPU produces 1.7 n words in microseconds);
  //   2. the draws j_i of the Fisher-Yates steps i = n-1 .. 1 are read off it in stream order — a rejected word shifts
  //      every later draw, so this scan is sequential, but branch-free: the store is unconditional, the step index
  //      advances by the acceptance bit, and the shift of the rejection zone is constant between powers of two;
  //   3. the final occupant of a position is traced back through the swaps in reverse time (i = 1 .. n-1): position p
  //      came from j_i if p == i, from i if p == j_i.  Only `take` positions are followed: one bit per position says
  //      whether it is followed (n / 8 bytes, cache resident), and about take * ln(n / take) swaps hit one.
  // Returns false if the keystream source failed (the generator is then unchanged).
  bool shuffle_head(uint64_t n, uint64_t take, uint32_t *out, KeystreamSource &ks, std::vector<uint32_t> &draws,
                    std::vector<uint64_t> &bits) {
    if (take > n) take = n;
    if (n > 0xFFFFFFFFull) {  // (gen_index samples 64-bit words above 2^32: plain shuffle)
      std::vector<uint64_t> v(n);
      for (uint64_t i = 0; i < n; ++i) v[i] = i;
      shuffle(v.data(), n);
      for (uint64_t t = 0; t < take; ++t) out[t] = (uint32_t)v[t];
      return true;
    }
    draws.resize(n > 0 ? n : 1);
    const uint64_t start = abs_pos();
    uint64_t pos = start;                 // next keystream word to read
    uint64_t i = n > 0 ? n - 1 : 0;       // current step: j_i is drawn from 0..=i
    uint64_t chunk_blocks = (n + n / 2 + n / 4) / 16 + 64;
    while (i >= 1) {
      const uint64_t first_block = pos / 16;
      const uint32_t *w = ks.blocks(key_, first_block, chunk_blocks);
      if (!w) return false;
      const uint32_t *wp = w + (pos - first_block * 16), *const wend = w + chunk_blocks * 16;
      while (i >= 1 && wp < wend) {
        const uint32_t s = (uint32_t)__builtin_clz((uint32_t)(i + 1));
        const uint64_t band_lo = (0x80000000ull >> s) - 1 > 1 ? (0x80000000ull >> s) - 1 : 1;  // range >= 2^(31-s)
        uint64_t ii = i;
        // two words per round: the second word's product is formed for both outcomes of the first (same range, or
        // range - 1) and selected, which halves the length of the dependent chain per word
        while (ii > band_lo && wp + 1 < wend) {
          const uint32_t r0 = (uint32_t)(ii + 1);
          const uint32_t v0 = wp[0], v1 = wp[1];
          wp += 2;
          // v1 * (r0 - 1) = v1 * r0 - v1 and ((r0 - 1) << s) - 1 = z0 - (1 << s): the selection is arithmetic, not a branch
          const uint64_t m0 = (uint64_t)v0 * r0, m1a = (uint64_t)v1 * r0;
          const uint32_t z0 = (r0 << s) - 1u;
          const uint32_t acc0 = (uint32_t)m0 <= z0;
          const uint64_t m1 = m1a - (uint64_t)(v1 & (0u - acc0));
          const uint32_t acc1 = (uint32_t)m1 <= z0 - (acc0 << s);
          draws[ii] = (uint32_t)(m0 >> 32);
          ii -= acc0;
          draws[ii] = (uint32_t)(m1 >> 32);  // (overwrites the first word's value if that one was rejected)
          ii -= acc1;
        }
        while (ii >= band_lo && wp < wend) {
          const uint32_t range = (uint32_t)(ii + 1);
          const uint64_t m = (uint64_t)(*wp++) * range;
          draws[ii] = (uint32_t)(m >> 32);
          ii -= (uint32_t)m <= (range << s) - 1u;  // accepted: next step
        }
        i = ii;
      }
      pos = first_block * 16 + (uint64_t)(wp - w);
      chunk_blocks = chunk_blocks / 8 + 64;
    }
    seek_abs(pos);
    bits.assign((n + 63) / 64, 0);
    constexpr uint64_t kSieve = 1u << 17;      // bits of a never-cleared pre-filter on j that stays in L1 (16 KB)
    std::vector<uint64_t> sieve(kSieve / 64, 0);
    std::vector<uint32_t> where(take);  // where[t] = where the final occupant of position t sits at the current time
    auto follow = [&](uint64_t p) { bits[p >> 6] |= 1ull << (p & 63); sieve[(p & (kSieve - 1)) >> 6] |= 1ull << (p & 63); };
    for (uint64_t t = 0; t < take; ++t) { where[t] = (uint32_t)t; follow(t); }
    auto slot_of = [&](uint32_t p) { for (uint64_t t = 0; t < take; ++t) if (where[t] == p) return t; return take; };
    for (uint64_t k = 1; k < n; ++k) {
      const uint32_t j = draws[k];
      const bool at_k = (bits[k >> 6] >> (k & 63)) & 1;                              // (sequential reads)
      const bool maybe_j = (sieve[(j & (kSieve - 1)) >> 6] >> (j & 63)) & 1;
      if (!(at_k | maybe_j)) continue;
      const bool at_j = (bits[j >> 6] >> (j & 63)) & 1;
      if (!(at_k | at_j) || j == k) continue;
      const uint64_t a = at_k ? slot_of((uint32_t)k) : take, b = at_j ? slot_of(j) : take;
      if (a < take) where[a] = j;
      if (b < take) where[b] = (uint32_t)k;
      if (at_k != at_j) {  // one followed position moved: k <-> j
        if (at_k) { bits[k >> 6] ^= 1ull << (k & 63); follow(j); }
        else { bits[j >> 6] ^= 1ull << (j & 63); follow(k); }
      }
    }
    for (uint64_t t = 0; t < take; ++t) out[t] = where[t];  // the identity permutation holds value p at position p
    return true;
  }

  // position in the keystream, in 32-bit words since seeding, and the way back
  uint64_t abs_pos() const { return index_ >= 64 ? counter_ * 16 : (counter_ - 4) * 16 + index_; }
  void seek_abs(uint64_t word_pos) {
    counter_ = (word_pos / 64) * 4;
    index_ = 64;
    if (word_pos % 64) { refill(); index_ = (uint32_t)(word_pos % 64); }
  }
  // blocks [first, first + n) of this generator's keystream, 16 words each (what a KeystreamSource must produce)
  static void keystream_blocks(const uint32_t key[8], uint64_t first, uint64_t n, uint32_t *out) {
    StdRng g;
    std::memcpy(g.key_, key, sizeof(g.key_));
    uint64_t b = 0;
    for (; b + 4 <= n; b += 4) { g.counter_ = first + b; g.refill(); std::memcpy(out + 16 * b, g.buf_, sizeof(g.buf_)); }
    for (; b < n; ++b) g.block(first + b, out + 16 * b);
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
    return weighted_index_cum(cum, total, n);
  }

  // ... with the cumulative weights already formed: cum[i] = w[0] + .. + w[i] summed left to right (i < n - 1),
  // total = the same sum over all n
  uint64_t weighted_index_cum(const float *cum, float total, uint64_t n) {
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
  StdRng() : counter_(0), index_(64) {}
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

  // the four blocks of a refill in the four lanes of a vector (the block function is the same for every counter)
  typedef uint32_t u32x4 __attribute__((vector_size(16)));
  static u32x4 rotl4(u32x4 x, int n) { return (x << n) | (x >> (32 - n)); }
  void refill() {
    u32x4 s[16], x[16];
    const uint32_t c0[4] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u};
    for (int i = 0; i < 4; ++i) s[i] = u32x4{c0[i], c0[i], c0[i], c0[i]};
    for (int i = 0; i < 8; ++i) s[4 + i] = u32x4{key_[i], key_[i], key_[i], key_[i]};
    const uint64_t c = counter_;
    s[12] = u32x4{(uint32_t)c, (uint32_t)(c + 1), (uint32_t)(c + 2), (uint32_t)(c + 3)};
    s[13] = u32x4{(uint32_t)(c >> 32), (uint32_t)((c + 1) >> 32), (uint32_t)((c + 2) >> 32), (uint32_t)((c + 3) >> 32)};
    s[14] = u32x4{0, 0, 0, 0};
    s[15] = u32x4{0, 0, 0, 0};
    for (int i = 0; i < 16; ++i) x[i] = s[i];
#define VI_QR(a, b, c, d)                                   \
    x[a] += x[b]; x[d] = rotl4(x[d] ^ x[a], 16);             \
    x[c] += x[d]; x[b] = rotl4(x[b] ^ x[c], 12);             \
    x[a] += x[b]; x[d] = rotl4(x[d] ^ x[a], 8);              \
    x[c] += x[d]; x[b] = rotl4(x[b] ^ x[c], 7);
    for (int r = 0; r < 6; ++r) {  // 12 rounds
      VI_QR(0, 4, 8, 12) VI_QR(1, 5, 9, 13) VI_QR(2, 6, 10, 14) VI_QR(3, 7, 11, 15)
      VI_QR(0, 5, 10, 15) VI_QR(1, 6, 11, 12) VI_QR(2, 7, 8, 13) VI_QR(3, 4, 9, 14)
    }
#undef VI_QR
    for (int i = 0; i < 16; ++i) {
      const u32x4 o = x[i] + s[i];
      for (int b = 0; b < 4; ++b) buf_[16 * b + i] = o[b];
    }
    counter_ += 4;
  }

  uint32_t key_[8];
  uint64_t counter_;
  uint32_t buf_[64];
  uint32_t index_;
};

struct HostKeystream : KeystreamSource {
  std::vector<uint32_t> buf;
  const uint32_t *blocks(const uint32_t key[8], uint64_t first_block, uint64_t nblocks) override {
    buf.resize(nblocks * 16);
    StdRng::keystream_blocks(key, first_block, nblocks, buf.data());
    return buf.data();
  }
};

}  // namespace vi
