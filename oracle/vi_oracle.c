/*
 * vi_oracle.c — CPU ORACLE (test infrastructure, NOT product code).
 * See vi_oracle.h for scope, citations and the parity pin status.
 *
 * Build: gcc -O2 -ffp-contract=off -fopenmp -fPIC -shared (oracle/Makefile).
 * -ffp-contract=off is REQUIRED: the reference (Rust) never fuses a*b+c.
 */
#define _GNU_SOURCE
#include "vi_oracle.h"
#include "../include/vi_reduce_order.h"

#include <errno.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------- */
/* heuristics                                                                 */
/* ------------------------------------------------------------------------- */

/* src/utils.rs:9-16 */
uint64_t orc_calculate_num_clusters(uint64_t n) {
  if (n < 10000) return (uint64_t)sqrt((double)n);
  if (n < 100000) return 2 * (uint64_t)ceil(sqrt((double)n));
  return 4 * (uint64_t)ceil(sqrt((double)n));
}

/* src/utils.rs:18-26 */
uint64_t orc_calculate_max_iterations(uint64_t n) {
  if (n < 10000) return 300;
  if (n < 100000) return 100;
  if (n < 1000000) return 50;
  return 20;
}

/* src/kmeans.rs:83  min(256, max(10, (n as f32).sqrt() as usize)) */
uint64_t orc_minibatch_size(uint64_t n) {
  uint64_t s = (uint64_t)sqrtf((float)n);
  if (s < 10) s = 10;
  if (s > 256) s = 256;
  return s;
}

/* src/kmeans.rs:483  ((k as f32).sqrt() as usize).max(2).min(k / 2) */
uint64_t orc_meta_k(uint64_t k) {
  uint64_t m = (uint64_t)sqrtf((float)k);
  if (m < 2) m = 2;
  if (m > k / 2) m = k / 2;
  return m;
}

/* src/ivf_index.rs:104  (k as f32).sqrt().ceil() as usize */
uint64_t orc_num_shards(uint64_t k) { return (uint64_t)ceilf(sqrtf((float)k)); }

/* ------------------------------------------------------------------------- */
/* distances                                                                  */
/* ------------------------------------------------------------------------- */

/* src/utils.rs:28-30 */
float orc_l2sq_scalar(const float *a, const float *b, size_t d) {
  float acc = 0.0f;
  for (size_t j = 0; j < d; ++j) {
    float t = a[j] - b[j];
    acc = acc + t * t;
  }
  return acc;
}

/* src/kmeans.rs:377-419.  wide 0.7.33 reduce_add order: see header. */
float orc_l2sq_simd(const float *p, const float *c, size_t d) {
  float a8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  float a4[4] = {0, 0, 0, 0};
  size_t j = 0;
  while (j + 8 <= d) {
    for (int l = 0; l < 8; ++l) {
      float t = p[j + l] - c[j + l];
      a8[l] = a8[l] + t * t;
    }
    j += 8;
  }
  while (j + 4 <= d) {
    for (int l = 0; l < 4; ++l) {
      float t = p[j + l] - c[j + l];
      a4[l] = a4[l] + t * t;
    }
    j += 4;
  }
  float tail = 0.0f;
  while (j < d) {
    float t = p[j] - c[j];
    tail = tail + t * t;
    j += 1;
  }
  /* lane order of wide's reduce_add: include/vi_reduce_order.h (shared with the product; unpinned) */
  float lo = VI_REDUCE4(a8[0], a8[1], a8[2], a8[3]);
  float hi = VI_REDUCE4(a8[4], a8[5], a8[6], a8[7]);
  float r8 = lo + hi;
  float r4 = VI_REDUCE4(a4[0], a4[1], a4[2], a4[3]);
  return (r8 + r4) + tail;
}

/* ------------------------------------------------------------------------- */
/* rand 0.8.5 StdRng (ChaCha12) restatement                                   */
/* ------------------------------------------------------------------------- */

static inline uint32_t rotl32(uint32_t x, int n) { return (x << n) | (x >> (32 - n)); }

#define QR(a, b, c, d)                                                              \
  do {                                                                              \
    a += b; d ^= a; d = rotl32(d, 16);                                              \
    c += d; b ^= c; b = rotl32(b, 12);                                              \
    a += b; d ^= a; d = rotl32(d, 8);                                               \
    c += d; b ^= c; b = rotl32(b, 7);                                               \
  } while (0)

void orc_chacha_block(const uint32_t key[8], uint64_t counter, uint64_t stream,
                      int rounds, uint32_t out[16]) {
  uint32_t s[16], x[16];
  s[0] = 0x61707865u; s[1] = 0x3320646eu; s[2] = 0x79622d32u; s[3] = 0x6b206574u;
  for (int i = 0; i < 8; ++i) s[4 + i] = key[i];
  s[12] = (uint32_t)counter; s[13] = (uint32_t)(counter >> 32);
  s[14] = (uint32_t)stream;  s[15] = (uint32_t)(stream >> 32);
  memcpy(x, s, sizeof(s));
  for (int r = 0; r < rounds; r += 2) {
    QR(x[0], x[4], x[8], x[12]); QR(x[1], x[5], x[9], x[13]);
    QR(x[2], x[6], x[10], x[14]); QR(x[3], x[7], x[11], x[15]);
    QR(x[0], x[5], x[10], x[15]); QR(x[1], x[6], x[11], x[12]);
    QR(x[2], x[7], x[8], x[13]); QR(x[3], x[4], x[9], x[14]);
  }
  for (int i = 0; i < 16; ++i) out[i] = x[i] + s[i];
}

/* rand_chacha fills a 4-block (64 word) buffer per refill */
static void rng_refill(orc_rng *r) {
  for (int b = 0; b < 4; ++b) orc_chacha_block(r->key, r->counter + b, 0, 12, r->buf + 16 * b);
  r->counter += 4;
}

void orc_rng_from_seed(orc_rng *r, const uint8_t seed[32]) {
  for (int i = 0; i < 8; ++i)
    r->key[i] = (uint32_t)seed[4 * i] | ((uint32_t)seed[4 * i + 1] << 8) |
                ((uint32_t)seed[4 * i + 2] << 16) | ((uint32_t)seed[4 * i + 3] << 24);
  r->counter = 0;
  r->index = 64; /* empty buffer */
}

/* rand_core 0.6.4 SeedableRng::seed_from_u64: PCG32 expansion of the u64 */
void orc_rng_seed_from_u64(orc_rng *r, uint64_t state) {
  const uint64_t MUL = 6364136223846793005ULL, INC = 11634580027462260723ULL;
  uint8_t seed[32];
  for (int i = 0; i < 8; ++i) {
    state = state * MUL + INC;
    uint32_t xorshifted = (uint32_t)(((state >> 18) ^ state) >> 27);
    uint32_t rot = (uint32_t)(state >> 59);
    uint32_t x = (xorshifted >> rot) | (xorshifted << ((32 - rot) & 31));
    seed[4 * i] = (uint8_t)x; seed[4 * i + 1] = (uint8_t)(x >> 8);
    seed[4 * i + 2] = (uint8_t)(x >> 16); seed[4 * i + 3] = (uint8_t)(x >> 24);
  }
  orc_rng_from_seed(r, seed);
}

/* rand_core BlockRng::next_u32 */
uint32_t orc_rng_next_u32(orc_rng *r) {
  if (r->index >= 64) { rng_refill(r); r->index = 0; }
  return r->buf[r->index++];
}

/* rand_core BlockRng::next_u64 (three cases, incl. the buffer-straddling one) */
uint64_t orc_rng_next_u64(orc_rng *r) {
  uint32_t idx = r->index;
  if (idx < 63) {
    r->index += 2;
    return ((uint64_t)r->buf[idx + 1] << 32) | r->buf[idx];
  } else if (idx >= 64) {
    rng_refill(r);
    r->index = 2;
    return ((uint64_t)r->buf[1] << 32) | r->buf[0];
  } else {
    uint64_t x = r->buf[63];
    rng_refill(r);
    r->index = 1;
    return ((uint64_t)r->buf[0] << 32) | x;
  }
}

static inline int clz64(uint64_t x) { return x ? __builtin_clzll(x) : 64; }
static inline int clz32(uint32_t x) { return x ? __builtin_clz(x) : 32; }

/* rand 0.8.5 UniformInt<usize>::sample_single_inclusive via sample_single */
uint64_t orc_rng_gen_range_usize(orc_rng *r, uint64_t low, uint64_t high) {
  uint64_t range = (high - 1) - low + 1; /* high > low asserted by rand */
  if (range == 0) return orc_rng_next_u64(r);
  uint64_t zone = (range << clz64(range)) - 1;
  for (;;) {
    uint64_t v = orc_rng_next_u64(r);
    unsigned __int128 m = (unsigned __int128)v * range;
    uint64_t hi = (uint64_t)(m >> 64), lo = (uint64_t)m;
    if (lo <= zone) return low + hi;
  }
}

static uint32_t rng_gen_range_u32(orc_rng *r, uint32_t low, uint32_t high) {
  uint32_t range = (high - 1) - low + 1;
  if (range == 0) return orc_rng_next_u32(r);
  uint32_t zone = (range << clz32(range)) - 1;
  for (;;) {
    uint32_t v = orc_rng_next_u32(r);
    uint64_t m = (uint64_t)v * range;
    uint32_t hi = (uint32_t)(m >> 32), lo = (uint32_t)m;
    if (lo <= zone) return low + hi;
  }
}

/* rand 0.8.5 seq::gen_index */
static uint64_t rng_gen_index(orc_rng *r, uint64_t ubound) {
  if (ubound <= 0xFFFFFFFFull) return rng_gen_range_u32(r, 0, (uint32_t)ubound);
  return orc_rng_gen_range_usize(r, 0, ubound);
}

/* SliceRandom::shuffle */
void orc_rng_shuffle_u64(orc_rng *r, uint64_t *v, uint64_t n) {
  for (uint64_t i = n; i-- > 1;) {
    uint64_t j = rng_gen_index(r, i + 1);
    uint64_t t = v[i]; v[i] = v[j]; v[j] = t;
  }
}

/* IteratorRandom::choose_multiple over 0..n */
static void rng_choose_multiple_range(orc_rng *r, uint64_t n, uint64_t amount, uint64_t *out,
                                      uint64_t *out_len) {
  uint64_t len = amount < n ? amount : n;
  for (uint64_t i = 0; i < len; ++i) out[i] = i;
  *out_len = len;
  if (len == amount) {
    for (uint64_t i = 0; amount + i < n; ++i) {
      uint64_t k = rng_gen_index(r, i + 1 + amount);
      if (k < amount) out[k] = amount + i;
    }
  }
}

static inline float f32_from_bits(uint32_t b) { float f; memcpy(&f, &b, 4); return f; }
static inline uint32_t f32_bits(float f) { uint32_t b; memcpy(&b, &f, 4); return b; }

/* rand 0.8.5 WeightedIndex<f32>::new + sample.  cum has n-1 entries. */
static uint64_t rng_weighted_index_sample(orc_rng *r, const float *w, uint64_t n, float *cum) {
  float total = w[0];
  for (uint64_t i = 1; i < n; ++i) { cum[i - 1] = total; total += w[i]; }
  /* UniformFloat::<f32>::new(0, total) */
  float low = 0.0f, high = total;
  float max_rand = f32_from_bits((127u << 23) | 0x7FFFFFu) - 1.0f;
  float scale = high - low;
  while (!(scale * max_rand + low < high)) scale = f32_from_bits(f32_bits(scale) - 1);
  /* sample */
  float v12 = f32_from_bits((127u << 23) | (orc_rng_next_u32(r) >> 9));
  float v01 = v12 - 1.0f;
  float chosen = v01 * scale + low;
  /* partition point: first cum[i] with !(cum[i] <= chosen) */
  uint64_t lo = 0, hi = n - 1;
  while (lo < hi) {
    uint64_t mid = lo + (hi - lo) / 2;
    if (cum[mid] <= chosen) lo = mid + 1; else hi = mid;
  }
  return lo;
}

/* ------------------------------------------------------------------------- */
/* k-means                                                                    */
/* ------------------------------------------------------------------------- */

/* src/kmeans.rs:355-373 */
void orc_find_nearest_centroid(const float *p, const float *C, size_t k, size_t d,
                               uint64_t *best, float *best_dist) {
  uint64_t bc = 0;
  float bd = INFINITY;
  for (size_t i = 0; i < k; ++i) {
    float dist = orc_l2sq_simd(p, C + i * d, d);
    if (dist < bd) { bd = dist; bc = i; }
  }
  *best = bc;
  if (best_dist) *best_dist = bd;
}

/* src/kmeans.rs:462-470 */
void orc_assign_brute_force(const float *X, size_t n, size_t d, const float *C, size_t k,
                            uint64_t *labels) {
#pragma omp parallel for schedule(static)
  for (long i = 0; i < (long)n; ++i)
    orc_find_nearest_centroid(X + (size_t)i * d, C, k, d, &labels[i], NULL);
}

/* src/kmeans.rs:584-648 */
void orc_build_centroid_hierarchy(const float *C, size_t k, size_t d, size_t meta_k,
                                  uint64_t seed, float *meta, uint64_t *c2m) {
  orc_rng rng;
  orc_rng_seed_from_u64(&rng, seed);
  uint64_t *chosen = (uint64_t *)malloc(sizeof(uint64_t) * (meta_k ? meta_k : 1));
  uint64_t nchosen = 0;
  rng_choose_multiple_range(&rng, k, meta_k, chosen, &nchosen);
  memset(meta, 0, sizeof(float) * meta_k * d);
  for (uint64_t i = 0; i < nchosen; ++i) memcpy(meta + i * d, C + chosen[i] * d, sizeof(float) * d);
  free(chosen);
  memset(c2m, 0, sizeof(uint64_t) * k);
  float *sum = (float *)malloc(sizeof(float) * d);
  for (int iter = 0; iter < 5; ++iter) {
    for (size_t c = 0; c < k; ++c) {
      uint64_t bm = 0;
      float bd = INFINITY;
      for (size_t m = 0; m < meta_k; ++m) {
        float dist = orc_l2sq_simd(C + c * d, meta + m * d, d);
        if (dist < bd) { bd = dist; bm = m; }
      }
      c2m[c] = bm;
    }
    for (size_t m = 0; m < meta_k; ++m) {
      size_t count = 0;
      for (size_t j = 0; j < d; ++j) sum[j] = 0.0f;
      for (size_t c = 0; c < k; ++c)
        if (c2m[c] == m) {
          count += 1;
          for (size_t j = 0; j < d; ++j) sum[j] += C[c * d + j];
        }
      if (count > 0)
        for (size_t j = 0; j < d; ++j) meta[m * d + j] = sum[j] / (float)count;
    }
  }
  free(sum);
}

typedef struct { float dist; uint32_t idx; } dist_idx;
/* stable ordering == sort by (dist, original position) */
static int cmp_dist_idx(const void *a, const void *b) {
  const dist_idx *x = (const dist_idx *)a, *y = (const dist_idx *)b;
  if (x->dist < y->dist) return -1;
  if (x->dist > y->dist) return 1;
  return (x->idx > y->idx) - (x->idx < y->idx);
}

/* src/kmeans.rs:474-581 */
void orc_assign_hierarchical(const float *X, size_t n, size_t d, const float *C, size_t k,
                             uint64_t seed, uint64_t *labels) {
  size_t meta_k = orc_meta_k(k);
  uint64_t hseed = seed * 17ULL + 42ULL; /* wrapping */
  float *meta = (float *)malloc(sizeof(float) * meta_k * d);
  uint64_t *c2m = (uint64_t *)malloc(sizeof(uint64_t) * k);
  orc_build_centroid_hierarchy(C, k, d, meta_k, hseed, meta, c2m);
  /* meta_to_centroids: CSR, ascending c within a meta cluster (:518-521) */
  uint64_t *off = (uint64_t *)calloc(meta_k + 1, sizeof(uint64_t));
  for (size_t c = 0; c < k; ++c) off[c2m[c] + 1]++;
  for (size_t m = 0; m < meta_k; ++m) off[m + 1] += off[m];
  uint64_t *members = (uint64_t *)malloc(sizeof(uint64_t) * (k ? k : 1));
  uint64_t *cur = (uint64_t *)malloc(sizeof(uint64_t) * (meta_k + 1));
  memcpy(cur, off, sizeof(uint64_t) * (meta_k + 1));
  for (size_t c = 0; c < k; ++c) members[cur[c2m[c]]++] = c;
  free(cur);
  size_t topk = meta_k < 3 ? meta_k : 3;
#pragma omp parallel
  {
    dist_idx *md = (dist_idx *)malloc(sizeof(dist_idx) * meta_k);
#pragma omp for schedule(static)
    for (long i = 0; i < (long)n; ++i) {
      const float *p = X + (size_t)i * d;
      for (size_t m = 0; m < meta_k; ++m) {
        md[m].dist = orc_l2sq_simd(p, meta + m * d, d);
        md[m].idx = (uint32_t)m;
      }
      qsort(md, meta_k, sizeof(dist_idx), cmp_dist_idx); /* :666 stable sort */
      /* candidates in (meta rank, ascending c) order; find_nearest over them
       * with strict '<' (:553) */
      uint64_t best = 0;
      float bd = INFINITY;
      int first = 1;
      for (size_t t = 0; t < topk; ++t) {
        uint64_t m = md[t].idx;
        for (uint64_t e = off[m]; e < off[m + 1]; ++e) {
          uint64_t c = members[e];
          float dist = orc_l2sq_simd(p, C + c * d, d);
          if (first) { best = c; first = 0; } /* best_c = 0 => first candidate */
          if (dist < bd) { bd = dist; best = c; }
        }
      }
      labels[i] = best;
    }
    free(md);
  }
  free(meta); free(c2m); free(off); free(members);
}

/* src/kmeans.rs:445-459 */
void orc_assign(const float *X, size_t n, size_t d, const float *C, size_t k, uint64_t seed,
                uint64_t *labels) {
  if (k > 100) orc_assign_hierarchical(X, n, d, C, k, seed, labels);
  else orc_assign_brute_force(X, n, d, C, k, labels);
}

/* src/kmeans.rs:422-443: rows 0..m of X against one centroid */
static void update_min_distances(const float *X, size_t d, const float *c, float *min_d, size_t m) {
#pragma omp parallel for schedule(static)
  for (long i = 0; i < (long)m; ++i) {
    float dist = orc_l2sq_simd(X + (size_t)i * d, c, d);
    if (dist < min_d[i]) min_d[i] = dist;
  }
}

/* src/kmeans.rs:154-310 */
void orc_kmeans_pp_init(const float *X, size_t n, size_t d, size_t k, uint64_t seed, float *C) {
  const size_t sample_threshold = 50000;
  orc_rng rng;
  orc_rng_seed_from_u64(&rng, seed);
  size_t actual_k = k < n ? k : n;
  memset(C, 0, sizeof(float) * k * d);
  uint64_t first = orc_rng_gen_range_usize(&rng, 0, n);
  memcpy(C, X + first * d, sizeof(float) * d);
  int sampled = n > sample_threshold;
  uint64_t *sample_idx = NULL;
  size_t m = n;
  if (sampled) {
    sample_idx = (uint64_t *)malloc(sizeof(uint64_t) * n);
    for (size_t i = 0; i < n; ++i) sample_idx[i] = i;
    orc_rng_shuffle_u64(&rng, sample_idx, n);
    m = sample_threshold < n ? sample_threshold : n;
  }
  float *min_d = (float *)malloc(sizeof(float) * m);
  float *w = (float *)malloc(sizeof(float) * m);
  float *cum = (float *)malloc(sizeof(float) * m);
  for (size_t i = 0; i < m; ++i) min_d[i] = INFINITY;
  for (size_t i = 1; i < actual_k; ++i) {
    /* NB (:268,:435): the sampled variant measures rows 0..m, not sample_idx */
    update_min_distances(X, d, C + (i - 1) * d, min_d, m);
    float total = 0.0f;
    for (size_t j = 0; j < m; ++j) { w[j] = min_d[j] * min_d[j]; total += w[j]; }
    if (total == 0.0f) {
      uint64_t dup = orc_rng_gen_range_usize(&rng, 0, i);
      memcpy(C + i * d, C + dup * d, sizeof(float) * d);
    } else {
      uint64_t s = rng_weighted_index_sample(&rng, w, m, cum);
      uint64_t chosen = sampled ? sample_idx[s] : s;
      memcpy(C + i * d, X + chosen * d, sizeof(float) * d);
    }
  }
  for (size_t i = actual_k; i < k; ++i) {
    uint64_t dup = orc_rng_gen_range_usize(&rng, 0, actual_k);
    memcpy(C + i * d, C + dup * d, sizeof(float) * d);
  }
  free(min_d); free(w); free(cum); free(sample_idx);
}

/* src/kmeans.rs:313-331 */
static void handle_empty_clusters(float *C, const uint64_t *counts, size_t k, size_t d,
                                  const float *X, size_t n, orc_rng *rng) {
  for (size_t c = 0; c < k; ++c)
    if (counts[c] == 0) {
      uint64_t ri = orc_rng_gen_range_usize(rng, 0, n);
      memcpy(C + c * d, X + ri * d, sizeof(float) * d);
    }
}

/* src/kmeans.rs:334-351 (Rayon's sum order is unspecified; sequential here) */
static float centroid_delta(const float *cur, const float *prev, size_t k, size_t d) {
  float dsq = 0.0f;
  for (size_t c = 0; c < k; ++c) {
    float local = 0.0f;
    for (size_t j = 0; j < d; ++j) {
      float diff = cur[c * d + j] - prev[c * d + j];
      local += diff * diff;
    }
    dsq += local;
  }
  return sqrtf(dsq / (float)(k * d));
}

/* src/kmeans.rs:674-719 */
void orc_update_centroids(const float *X, size_t n, size_t d, const uint64_t *labels, size_t k,
                          float *C_new, uint64_t *counts) {
  memset(C_new, 0, sizeof(float) * k * d);
  memset(counts, 0, sizeof(uint64_t) * k);
  for (size_t i = 0; i < n; ++i) {
    uint64_t c = labels[i];
    counts[c] += 1;
    for (size_t j = 0; j < d; ++j) C_new[c * d + j] += X[i * d + j];
  }
  for (size_t c = 0; c < k; ++c)
    if (counts[c] > 0)
      for (size_t j = 0; j < d; ++j) C_new[c * d + j] /= (float)counts[c];
}

/* src/kmeans.rs:15-60 */
int orc_kmeans_parallel(const float *X, size_t n, size_t d, size_t k, size_t max_iters,
                        float thr, uint64_t seed, int force_brute, float *C,
                        uint64_t *labels, uint64_t *iters_run) {
  if (thr < 0) thr = 1e-4f;
  if (n == 0 || d == 0) return ORC_INVALID_INPUT;
  orc_rng rng;
  orc_rng_seed_from_u64(&rng, seed);
  orc_kmeans_pp_init(X, n, d, k, seed, C);
  memset(labels, 0, sizeof(uint64_t) * n);
  float *Cn = (float *)malloc(sizeof(float) * k * d);
  uint64_t *counts = (uint64_t *)malloc(sizeof(uint64_t) * k);
  uint64_t it = 0;
  for (; it < max_iters; ++it) {
    if (force_brute) orc_assign_brute_force(X, n, d, C, k, labels);
    else orc_assign(X, n, d, C, k, seed, labels);
    orc_update_centroids(X, n, d, labels, k, Cn, counts);
    handle_empty_clusters(Cn, counts, k, d, X, n, &rng);
    float delta = centroid_delta(Cn, C, k, d);
    memcpy(C, Cn, sizeof(float) * k * d);
    if (delta < thr) { ++it; break; }
  }
  if (iters_run) *iters_run = it;
  free(Cn); free(counts);
  return ORC_OK;
}

/* src/kmeans.rs:64-150 */
int orc_kmeans_mini_batch(const float *X, size_t n, size_t d, size_t k, size_t max_iters,
                          float thr, uint64_t seed, int force_brute, float *C,
                          uint64_t *labels, uint64_t *iters_run) {
  if (thr < 0) thr = 1e-4f;
  if (n == 0 || d == 0) return ORC_INVALID_INPUT;
  orc_rng rng;
  orc_rng_seed_from_u64(&rng, seed);
  size_t B = orc_minibatch_size(n);
  orc_kmeans_pp_init(X, n, d, k, seed, C);
  uint64_t *counts = (uint64_t *)calloc(k, sizeof(uint64_t));
  float *prev = (float *)malloc(sizeof(float) * k * d);
  memcpy(prev, C, sizeof(float) * k * d);
  uint64_t *perm = (uint64_t *)malloc(sizeof(uint64_t) * n);
  size_t nb = B < n ? B : n;
  uint64_t *bidx = (uint64_t *)malloc(sizeof(uint64_t) * nb);
  uint64_t *blab = (uint64_t *)malloc(sizeof(uint64_t) * nb);
  float *Ccur = (float *)malloc(sizeof(float) * k * d);
  float *bsum = (float *)malloc(sizeof(float) * d);
  uint64_t it = 0;
  for (; it < max_iters; ++it) {
    /* sample_batch :722-726 — full shuffle of 0..n, take B */
    for (size_t i = 0; i < n; ++i) perm[i] = i;
    orc_rng_shuffle_u64(&rng, perm, n);
    memcpy(bidx, perm, sizeof(uint64_t) * nb);
    /* batch assign :103-110 (always brute force over all k) */
#pragma omp parallel for schedule(static)
    for (long b = 0; b < (long)nb; ++b)
      orc_find_nearest_centroid(X + bidx[b] * d, C, k, d, &blab[b], NULL);
    /* update_centroids_mini_batch :729-787 */
    memcpy(Ccur, C, sizeof(float) * k * d);
    for (size_t c = 0; c < k; ++c) {
      size_t npts = 0;
      for (size_t j = 0; j < d; ++j) bsum[j] = 0.0f;
      for (size_t b = 0; b < nb; ++b)
        if (blab[b] == c) {
          npts += 1;
          for (size_t j = 0; j < d; ++j) bsum[j] += X[bidx[b] * d + j];
        }
      if (npts == 0) continue;
      uint64_t new_count = counts[c] + 1; /* per ITERATION, not per point */
      float eta = 1.0f / (float)new_count;
      for (size_t j = 0; j < d; ++j) {
        float mean = bsum[j] / (float)npts;
        C[c * d + j] = (1.0f - eta) * Ccur[c * d + j] + eta * mean;
      }
      counts[c] = new_count;
    }
    handle_empty_clusters(C, counts, k, d, X, n, &rng);
    float delta = centroid_delta(C, prev, k, d);
    memcpy(prev, C, sizeof(float) * k * d);
    if (delta < thr) { ++it; break; }
  }
  if (iters_run) *iters_run = it;
  /* labels == NULL: the caller checks the final assignment (:146-147) on sampled rows itself (tests at N = 1e7) */
  if (labels) {
    if (force_brute) orc_assign_brute_force(X, n, d, C, k, labels);
    else orc_assign(X, n, d, C, k, seed, labels);
  }
  free(counts); free(prev); free(perm); free(bidx); free(blab); free(Ccur); free(bsum);
  return ORC_OK;
}

/* ------------------------------------------------------------------------- */
/* shard files (src/shards.rs)                                                */
/* ------------------------------------------------------------------------- */

/* repr(C) layouts, src/shards.rs:22-51.  Header is 40 bytes (the "48" in the
 * reference comment is wrong: 8+8+4+4+8+8). */
typedef struct {
  uint64_t shard_id, version;
  uint32_t dimensions, num_centroids;
  uint64_t index_offset, data_offset;
} shard_header;
typedef struct {
  uint64_t centroid_id;
  uint32_t num_vectors, padding;
  uint64_t data_offset, data_size;
} centroid_index;
typedef struct { uint64_t id, external_id, timestamp; } vector_meta;

_Static_assert(sizeof(shard_header) == 40, "header");
_Static_assert(sizeof(centroid_index) == 32, "index entry");
_Static_assert(sizeof(vector_meta) == 24, "meta");

static int mkdir_p(const char *path) {
  char tmp[4096];
  size_t len = strlen(path);
  if (len == 0 || len >= sizeof(tmp)) return -1;
  memcpy(tmp, path, len + 1);
  for (size_t i = 1; i < len; ++i)
    if (tmp[i] == '/') {
      tmp[i] = 0;
      if (mkdir(tmp, 0777) != 0 && errno != EEXIST) return -1;
      tmp[i] = '/';
    }
  if (mkdir(tmp, 0777) != 0 && errno != EEXIST) return -1;
  return 0;
}

/* Shard::save_to src/shards.rs:68-177 */
int orc_shard_save_to(const char *shards_dir, uint64_t shard_id, uint32_t dim,
                      uint32_t num_lists, const uint64_t *centroid_ids,
                      const float *centroid_vecs, const uint64_t *list_off,
                      const uint64_t *ids, const uint64_t *ext_ids,
                      const uint64_t *timestamps, const float *vecs) {
  if (mkdir_p(shards_dir) != 0) return ORC_IO;
  char path[4096];
  snprintf(path, sizeof(path), "%s/shard_%llu.bin", shards_dir, (unsigned long long)shard_id);
  unlink(path);
  FILE *f = fopen(path, "wb");
  if (!f) return ORC_IO;
  size_t vsz = (size_t)dim * 4;
  size_t cpad = (8 - (vsz % 8)) % 8, vpad = cpad;
  shard_header h;
  h.shard_id = shard_id; h.version = 1; h.dimensions = dim; h.num_centroids = num_lists;
  h.index_offset = sizeof(shard_header);
  h.data_offset = sizeof(shard_header) + (uint64_t)num_lists * sizeof(centroid_index);
  int rc = ORC_OK;
  if (fwrite(&h, sizeof(h), 1, f) != 1) rc = ORC_IO;
  uint64_t cur = h.data_offset;
  for (uint32_t i = 0; i < num_lists && rc == ORC_OK; ++i) {
    uint64_t nv = list_off[i + 1] - list_off[i];
    centroid_index e;
    e.centroid_id = centroid_ids[i]; e.num_vectors = (uint32_t)nv; e.padding = 0;
    e.data_offset = cur;
    e.data_size = vsz + cpad + nv * (sizeof(vector_meta) + vsz + vpad);
    cur += e.data_size;
    if (fwrite(&e, sizeof(e), 1, f) != 1) rc = ORC_IO;
  }
  static const uint8_t zeros[8] = {0};
  for (uint32_t i = 0; i < num_lists && rc == ORC_OK; ++i) {
    if (fwrite(centroid_vecs + (size_t)i * dim, 1, vsz, f) != vsz) rc = ORC_IO;
    if (cpad && fwrite(zeros, 1, cpad, f) != cpad) rc = ORC_IO;
    for (uint64_t v = list_off[i]; v < list_off[i + 1] && rc == ORC_OK; ++v) {
      vector_meta m = {ids[v], ext_ids[v], timestamps[v]};
      if (fwrite(&m, sizeof(m), 1, f) != 1) rc = ORC_IO;
      if (fwrite(vecs + v * dim, 1, vsz, f) != vsz) rc = ORC_IO;
      if (vpad && fwrite(zeros, 1, vpad, f) != vpad) rc = ORC_IO;
    }
  }
  if (fclose(f) != 0) rc = ORC_IO;
  return rc;
}

static uint8_t *read_whole_file(const char *path, size_t *len) {
  FILE *f = fopen(path, "rb");
  if (!f) return NULL;
  fseek(f, 0, SEEK_END);
  long sz = ftell(f);
  fseek(f, 0, SEEK_SET);
  uint8_t *buf = (uint8_t *)malloc(sz > 0 ? (size_t)sz : 1);
  if (sz > 0 && fread(buf, 1, (size_t)sz, f) != (size_t)sz) { free(buf); fclose(f); return NULL; }
  fclose(f);
  *len = (size_t)sz;
  return buf;
}

/* Shard::get_centroid_vectors_from src/shards.rs:188-349 */
int orc_shard_get_centroid_vectors_from(const char *shards_dir, uint64_t shard_id,
                                        const uint64_t *centroid_ids, size_t n_req,
                                        uint32_t *dim_out, uint64_t *counts,
                                        float *centroid_out, uint64_t *metas_out,
                                        float *vecs_out) {
  char path[4096];
  snprintf(path, sizeof(path), "%s/shard_%llu.bin", shards_dir, (unsigned long long)shard_id);
  size_t len = 0;
  uint8_t *buf = read_whole_file(path, &len);
  if (!buf) return ORC_OTHER; /* :193-200 wraps the open error as Other */
  int rc = ORC_OK;
  shard_header h;
  if (len < sizeof(h)) { free(buf); return ORC_INVALID_DATA; }
  memcpy(&h, buf, sizeof(h));
  if (h.shard_id != shard_id) { free(buf); return ORC_INVALID_DATA; } /* :223-231 */
  size_t idx_bytes = (size_t)h.num_centroids * sizeof(centroid_index);
  if (h.index_offset > len || idx_bytes > len - h.index_offset) { free(buf); return ORC_INVALID_DATA; }
  const uint8_t *idx = buf + h.index_offset;
  size_t dims = h.dimensions, vsz = dims * 4;
  size_t cpad = (8 - (vsz % 8)) % 8, vpad = cpad;
  if (dim_out) *dim_out = h.dimensions;
  uint64_t vbase = 0;
  for (size_t r = 0; r < n_req && rc == ORC_OK; ++r) {
    centroid_index e;
    int found = 0;
    for (uint32_t i = 0; i < h.num_centroids; ++i) { /* linear find :257-265 */
      memcpy(&e, idx + (size_t)i * sizeof(e), sizeof(e));
      if (e.centroid_id == centroid_ids[r]) { found = 1; break; }
    }
    if (!found) { rc = ORC_NOT_FOUND; break; }
    if (e.data_offset > len || e.data_size > len - e.data_offset || e.data_size < vsz) {
      rc = ORC_OTHER; break; /* short read => "Failed to read cluster" */
    }
    const uint8_t *blk = buf + e.data_offset;
    if (counts) counts[r] = e.num_vectors;
    if (centroid_out) memcpy(centroid_out + r * dims, blk, vsz);
    size_t off = vsz + cpad;
    for (uint32_t v = 0; v < e.num_vectors; ++v) {
      if (off + sizeof(vector_meta) > e.data_size) { rc = ORC_INVALID_DATA; break; } /* :310-316 */
      if (off + sizeof(vector_meta) + vsz > e.data_size) { rc = ORC_INVALID_DATA; break; }
      if (metas_out) memcpy(metas_out + (vbase + v) * 3, blk + off, sizeof(vector_meta));
      off += sizeof(vector_meta);
      if (vecs_out) memcpy(vecs_out + (vbase + v) * dims, blk + off, vsz);
      off += vsz + vpad;
    }
    vbase += e.num_vectors;
  }
  free(buf);
  return rc;
}

/* ------------------------------------------------------------------------- */
/* index/index.bin codec: bincode 2 standard config over serde of
 * IvfIndex{centroids: Array1<Centroid>, centroids_to_shard: Array1<usize>,
 * dimension: u32} (src/ivf_index.rs:36-41,274-316).  PARITY UNPINNED.       */
/* ------------------------------------------------------------------------- */

typedef struct { uint8_t *p; size_t len, cap; } bytebuf;
static void bb_put(bytebuf *b, const void *src, size_t n) {
  if (b->len + n > b->cap) {
    size_t nc = b->cap ? b->cap * 2 : 4096;
    while (nc < b->len + n) nc *= 2;
    b->p = (uint8_t *)realloc(b->p, nc);
    b->cap = nc;
  }
  memcpy(b->p + b->len, src, n);
  b->len += n;
}
static void bb_varint(bytebuf *b, uint64_t v) {
  uint8_t t[9];
  if (v < 251) { t[0] = (uint8_t)v; bb_put(b, t, 1); }
  else if (v <= 0xFFFF) { t[0] = 251; t[1] = (uint8_t)v; t[2] = (uint8_t)(v >> 8); bb_put(b, t, 3); }
  else if (v <= 0xFFFFFFFFull) { t[0] = 252; for (int i = 0; i < 4; ++i) t[1 + i] = (uint8_t)(v >> (8 * i)); bb_put(b, t, 5); }
  else { t[0] = 253; for (int i = 0; i < 8; ++i) t[1 + i] = (uint8_t)(v >> (8 * i)); bb_put(b, t, 9); }
}
static int rd_varint(const uint8_t *p, size_t len, size_t *off, uint64_t *v) {
  if (*off >= len) return -1;
  uint8_t t = p[(*off)++];
  int nb;
  if (t < 251) { *v = t; return 0; }
  else if (t == 251) nb = 2; else if (t == 252) nb = 4; else if (t == 253) nb = 8; else return -1;
  if (*off + nb > len) return -1;
  uint64_t x = 0;
  for (int i = 0; i < nb; ++i) x |= (uint64_t)p[*off + i] << (8 * i);
  *off += nb;
  *v = x;
  return 0;
}

struct orc_index {
  uint32_t dim;
  uint64_t k;        /* non-empty centroids */
  float *C;          /* k x dim */
  uint64_t *c2s;     /* k */
  /* lists resident in RAM (the reference reads them from disk per query) */
  uint64_t *list_len;  /* k */
  uint64_t **list_meta; /* k -> 3*len u64 */
  float **list_vec;     /* k -> len*dim */
  uint8_t *list_ok;     /* 0 => shard unreadable: silently skipped (:253-254) */
};

static int index_save(const orc_index *ix, const char *index_dir) {
  if (mkdir_p(index_dir) != 0) return ORC_IO;
  bytebuf b = {0};
  uint8_t one = 1;
  bb_put(&b, &one, 1); bb_varint(&b, ix->k); bb_varint(&b, ix->k);
  for (uint64_t c = 0; c < ix->k; ++c) {
    bb_varint(&b, c); bb_varint(&b, ix->dim);
    bb_put(&b, ix->C + c * ix->dim, (size_t)ix->dim * 4);
  }
  bb_put(&b, &one, 1); bb_varint(&b, ix->k); bb_varint(&b, ix->k);
  for (uint64_t c = 0; c < ix->k; ++c) bb_varint(&b, ix->c2s[c]);
  bb_varint(&b, ix->dim);
  char path[4096];
  snprintf(path, sizeof(path), "%s/index.bin", index_dir);
  FILE *f = fopen(path, "wb");
  int rc = ORC_OK;
  if (!f) rc = ORC_IO;
  else {
    if (b.len && fwrite(b.p, 1, b.len, f) != b.len) rc = ORC_IO;
    if (fclose(f) != 0) rc = ORC_IO;
  }
  free(b.p);
  return rc;
}

static int index_load_bin(const char *index_dir, orc_index *ix) {
  char path[4096];
  snprintf(path, sizeof(path), "%s/index.bin", index_dir);
  size_t len = 0, off = 0;
  uint8_t *p = read_whole_file(path, &len);
  if (!p) return errno == ENOENT ? ORC_NOT_FOUND : ORC_IO;
  uint64_t v, k, k2;
  int rc = ORC_OTHER; /* bincode decode errors are wrapped as Other (:312) */
  if (off >= len || p[off++] != 1) goto done;
  if (rd_varint(p, len, &off, &k) || rd_varint(p, len, &off, &k2) || k != k2) goto done;
  ix->k = k;
  ix->C = NULL;
  uint64_t d0 = 0;
  for (uint64_t c = 0; c < k; ++c) {
    uint64_t id, dl;
    if (rd_varint(p, len, &off, &id) || rd_varint(p, len, &off, &dl)) goto done;
    if (c == 0) { d0 = dl; ix->C = (float *)malloc(sizeof(float) * (size_t)(k * dl + 1)); }
    if (dl != d0 || off + dl * 4 > len) goto done;
    memcpy(ix->C + c * d0, p + off, dl * 4);
    off += dl * 4;
  }
  if (off >= len || p[off++] != 1) goto done;
  if (rd_varint(p, len, &off, &v) || v != k || rd_varint(p, len, &off, &v) || v != k) goto done;
  ix->c2s = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(k + 1));
  for (uint64_t c = 0; c < k; ++c)
    if (rd_varint(p, len, &off, &ix->c2s[c])) goto done;
  if (rd_varint(p, len, &off, &v)) goto done;
  ix->dim = (uint32_t)v;
  if (k > 0 && d0 != ix->dim) goto done;
  rc = ORC_OK;
done:
  free(p);
  return rc;
}

/* Preload every list of every shard into RAM (one pass per shard file).  A shard that cannot be
 * read, has a mismatching id/dimension, or lacks one of its centroids leaves all its lists
 * unavailable — the reference drops the whole shard result in that case (:253-254, :257-265). */
static void index_load_lists(orc_index *ix, const char *shards_dir) {
  uint64_t k = ix->k;
  ix->list_len = (uint64_t *)calloc(k + 1, sizeof(uint64_t));
  ix->list_meta = (uint64_t **)calloc(k + 1, sizeof(uint64_t *));
  ix->list_vec = (float **)calloc(k + 1, sizeof(float *));
  ix->list_ok = (uint8_t *)calloc(k + 1, 1);
  uint64_t nshards = orc_index_num_shards(ix);
  size_t dims = ix->dim, vsz = dims * 4, cpad = (8 - (vsz % 8)) % 8;
  size_t stride = sizeof(vector_meta) + vsz + cpad;
  for (uint64_t s = 0; s < nshards; ++s) {
    char path[4096];
    snprintf(path, sizeof(path), "%s/shard_%llu.bin", shards_dir, (unsigned long long)s);
    size_t len = 0;
    uint8_t *buf = read_whole_file(path, &len);
    if (!buf) continue;
    shard_header h;
    int ok = len >= sizeof(h);
    if (ok) { memcpy(&h, buf, sizeof(h)); ok = h.shard_id == s && h.dimensions == ix->dim; }
    if (ok) ok = h.index_offset <= len && (size_t)h.num_centroids * sizeof(centroid_index) <= len - h.index_offset;
    /* first pass: every centroid of this shard must be present and in bounds */
    for (uint64_t c = 0; ok && c < k; ++c) {
      if (ix->c2s[c] != s) continue;
      int found = 0;
      for (uint32_t i = 0; i < h.num_centroids && !found; ++i) {
        centroid_index e;
        memcpy(&e, buf + h.index_offset + (size_t)i * sizeof(e), sizeof(e));
        if (e.centroid_id != c) continue;
        found = 1;
        size_t need = vsz + cpad + (size_t)e.num_vectors * stride - (e.num_vectors ? cpad : 0);
        if (e.data_offset > len || e.data_size > len - e.data_offset || need > e.data_size) ok = 0;
      }
      if (!found) ok = 0;
    }
    for (uint64_t c = 0; ok && c < k; ++c) {
      if (ix->c2s[c] != s) continue;
      for (uint32_t i = 0; i < h.num_centroids; ++i) {
        centroid_index e;
        memcpy(&e, buf + h.index_offset + (size_t)i * sizeof(e), sizeof(e));
        if (e.centroid_id != c) continue;
        uint64_t cnt = e.num_vectors;
        ix->list_meta[c] = (uint64_t *)malloc(sizeof(uint64_t) * 3 * (size_t)(cnt + 1));
        ix->list_vec[c] = (float *)malloc(sizeof(float) * (size_t)(cnt * dims + 1));
        const uint8_t *rec = buf + e.data_offset + vsz + cpad;
        for (uint64_t v = 0; v < cnt; ++v, rec += stride) {
          memcpy(ix->list_meta[c] + 3 * v, rec, sizeof(vector_meta));
          memcpy(ix->list_vec[c] + v * dims, rec + sizeof(vector_meta), vsz);
        }
        ix->list_len[c] = cnt;
        ix->list_ok[c] = 1;
        break;
      }
    }
    free(buf);
  }
}

int orc_index_load(const char *index_dir, const char *shards_dir, orc_index **out) {
  orc_index *ix = (orc_index *)calloc(1, sizeof(orc_index));
  int rc = index_load_bin(index_dir, ix);
  if (rc != ORC_OK) { orc_index_free(ix); return rc; }
  index_load_lists(ix, shards_dir);
  *out = ix;
  return ORC_OK;
}

void orc_index_free(orc_index *ix) {
  if (!ix) return;
  if (ix->list_meta) for (uint64_t c = 0; c < ix->k; ++c) free(ix->list_meta[c]);
  if (ix->list_vec) for (uint64_t c = 0; c < ix->k; ++c) free(ix->list_vec[c]);
  free(ix->list_meta); free(ix->list_vec); free(ix->list_len); free(ix->list_ok);
  free(ix->C); free(ix->c2s); free(ix);
}

uint64_t orc_index_num_centroids(const orc_index *ix) { return ix->k; }
uint32_t orc_index_dimension(const orc_index *ix) { return ix->dim; }
uint64_t orc_index_num_shards(const orc_index *ix) {
  uint64_t m = 0;
  for (uint64_t c = 0; c < ix->k; ++c) if (ix->c2s[c] + 1 > m) m = ix->c2s[c] + 1;
  return m;
}
void orc_index_centroids(const orc_index *ix, float *C_out, uint64_t *c2s_out) {
  if (C_out) memcpy(C_out, ix->C, sizeof(float) * ix->k * ix->dim);
  if (c2s_out) memcpy(c2s_out, ix->c2s, sizeof(uint64_t) * ix->k);
}
uint64_t orc_index_list_len(const orc_index *ix, uint64_t list) { return ix->list_len[list]; }

/* IvfIndex::fit_with_paths src/ivf_index.rs:58-177 + save_to :274-294 */
int orc_index_build(const float *X, const uint64_t *ext_ids, const uint64_t *timestamps,
                    size_t n, uint32_t dim, uint64_t nlist_override, uint64_t seed,
                    uint64_t now_secs, int force_brute, const char *index_dir,
                    const char *shards_dir, orc_index **out) {
  if (n == 0) return ORC_INVALID_INPUT; /* api.rs:116-118 */
  size_t d = dim;
  size_t k = nlist_override ? nlist_override : orc_calculate_num_clusters(n);
  size_t max_iters = orc_calculate_max_iterations(n);
  float *C = (float *)malloc(sizeof(float) * k * d);
  uint64_t *labels = (uint64_t *)malloc(sizeof(uint64_t) * n);
  int rc = orc_kmeans_mini_batch(X, n, d, k, max_iters, -1.0f, seed, force_brute, C, labels, NULL);
  if (rc != ORC_OK) { free(C); free(labels); return ORC_PANIC; }
  /* lists in ascending internal id (:94-101) */
  uint64_t *off = (uint64_t *)calloc(k + 1, sizeof(uint64_t));
  for (size_t i = 0; i < n; ++i) off[labels[i] + 1]++;
  for (size_t c = 0; c < k; ++c) off[c + 1] += off[c];
  uint64_t *order = (uint64_t *)malloc(sizeof(uint64_t) * n);
  {
    uint64_t *cur = (uint64_t *)malloc(sizeof(uint64_t) * (k + 1));
    memcpy(cur, off, sizeof(uint64_t) * (k + 1));
    for (size_t i = 0; i < n; ++i) order[cur[labels[i]]++] = i;
    free(cur);
  }
  /* super-centroids (:104-109) */
  size_t num_shards = orc_num_shards(k);
  uint64_t super_seed = seed * 31ULL + 7ULL;
  float *SC = (float *)malloc(sizeof(float) * num_shards * d);
  uint64_t *slab = (uint64_t *)malloc(sizeof(uint64_t) * k);
  rc = orc_kmeans_mini_batch(C, k, d, num_shards, 100, -1.0f, super_seed, force_brute, SC, slab, NULL);
  free(SC);
  if (rc != ORC_OK) { free(C); free(labels); free(off); free(order); free(slab); return ORC_PANIC; }
  /* drop empty lists, renumber (:123-164) */
  uint64_t *newid = (uint64_t *)malloc(sizeof(uint64_t) * k);
  size_t kk = 0;
  for (size_t c = 0; c < k; ++c) newid[c] = (off[c + 1] > off[c]) ? kk++ : (uint64_t)-1;
  orc_index *ix = (orc_index *)calloc(1, sizeof(orc_index));
  ix->dim = dim; ix->k = kk;
  ix->C = (float *)malloc(sizeof(float) * (kk * d + 1));
  ix->c2s = (uint64_t *)malloc(sizeof(uint64_t) * (kk + 1));
  for (size_t c = 0; c < k; ++c)
    if (newid[c] != (uint64_t)-1) {
      memcpy(ix->C + newid[c] * d, C + c * d, sizeof(float) * d);
      ix->c2s[newid[c]] = slab[c];
    }
  /* write every shard, including empty ones (:118-120,166-171) */
  int wrc = ORC_OK;
  for (size_t s = 0; s < num_shards; ++s) {
    size_t nl = 0, nv = 0;
    for (size_t c = 0; c < k; ++c)
      if (newid[c] != (uint64_t)-1 && slab[c] == s) { nl++; nv += off[c + 1] - off[c]; }
    uint64_t *cids = (uint64_t *)malloc(sizeof(uint64_t) * (nl + 1));
    float *cvec = (float *)malloc(sizeof(float) * (nl * d + 1));
    uint64_t *loff = (uint64_t *)malloc(sizeof(uint64_t) * (nl + 2));
    uint64_t *ids = (uint64_t *)malloc(sizeof(uint64_t) * (nv + 1));
    uint64_t *eids = (uint64_t *)malloc(sizeof(uint64_t) * (nv + 1));
    uint64_t *tss = (uint64_t *)malloc(sizeof(uint64_t) * (nv + 1));
    float *vv = (float *)malloc(sizeof(float) * (nv * d + 1));
    size_t li = 0, vi = 0;
    loff[0] = 0;
    for (size_t c = 0; c < k; ++c)
      if (newid[c] != (uint64_t)-1 && slab[c] == s) {
        cids[li] = newid[c];
        memcpy(cvec + li * d, C + c * d, sizeof(float) * d);
        for (uint64_t e = off[c]; e < off[c + 1]; ++e) {
          uint64_t i = order[e];
          ids[vi] = i; /* internal id = position (vector_store.rs:33) */
          eids[vi] = ext_ids ? ext_ids[i] : i;
          uint64_t ts = timestamps ? timestamps[i] : 0;
          tss[vi] = ts != 0 ? ts : now_secs;
          memcpy(vv + vi * d, X + i * d, sizeof(float) * d);
          vi++;
        }
        loff[++li] = vi;
      }
    int r = orc_shard_save_to(shards_dir, s, dim, (uint32_t)nl, cids, cvec, loff, ids, eids, tss, vv);
    if (r != ORC_OK) wrc = r; /* reference only eprintln!s (:168-170) */
    free(cids); free(cvec); free(loff); free(ids); free(eids); free(tss); free(vv);
  }
  (void)wrc;
  free(C); free(labels); free(off); free(order); free(slab); free(newid);
  rc = index_save(ix, index_dir);
  if (rc != ORC_OK) { orc_index_free(ix); return rc; }
  index_load_lists(ix, shards_dir);
  *out = ix;
  return ORC_OK;
}

/* coarse step src/ivf_index.rs:205-220 */
static int probe_lists(const orc_index *ix, const float *q, uint64_t n_probe, dist_idx *cd) {
  size_t d = ix->dim;
  for (uint64_t i = 0; i < ix->k; ++i) {
    cd[i].dist = orc_l2sq_scalar(q, ix->C + i * d, d);
    cd[i].idx = (uint32_t)i;
    if (cd[i].dist != cd[i].dist) return ORC_PANIC; /* partial_cmp().unwrap() */
  }
  qsort(cd, ix->k, sizeof(dist_idx), cmp_dist_idx);
  (void)n_probe;
  return ORC_OK;
}

int orc_index_probe(const orc_index *ix, const float *q, uint64_t n_probe, uint64_t *probes_out,
                    uint64_t *count) {
  if (n_probe == 0) return ORC_INVALID_INPUT;
  dist_idx *cd = (dist_idx *)malloc(sizeof(dist_idx) * (ix->k + 1));
  int rc = probe_lists(ix, q, n_probe, cd);
  uint64_t np = n_probe < ix->k ? n_probe : ix->k;
  if (rc == ORC_OK) {
    for (uint64_t i = 0; i < np; ++i) probes_out[i] = cd[i].idx;
    *count = np;
  }
  free(cd);
  return rc;
}

typedef struct { float dist; uint32_t order; uint64_t list; uint64_t pos; } cand;
static int cmp_cand(const void *a, const void *b) {
  const cand *x = (const cand *)a, *y = (const cand *)b;
  if (x->dist < y->dist) return -1;
  if (x->dist > y->dist) return 1;
  return (x->order > y->order) - (x->order < y->order);
}

/* IvfIndex::search_with_paths src/ivf_index.rs:190-267 */
int orc_index_search(const orc_index *ix, const float *q, uint64_t k, uint64_t n_probe,
                     uint64_t *ids_out, float *dist_out, float *vecs_out, uint64_t *count) {
  if (k == 0 || n_probe == 0) return ORC_INVALID_INPUT; /* :197-202 */
  size_t d = ix->dim;
  dist_idx *cd = (dist_idx *)malloc(sizeof(dist_idx) * (ix->k + 1));
  int rc = probe_lists(ix, q, n_probe, cd);
  if (rc != ORC_OK) { free(cd); return rc; }
  uint64_t np = n_probe < ix->k ? n_probe : ix->k;
  /* shard visiting order: first appearance in the probe list (one of the
   * reference's possible HashSet orders, :223-229) */
  uint64_t *shard_order = (uint64_t *)malloc(sizeof(uint64_t) * (np + 1));
  uint64_t ns = 0;
  for (uint64_t i = 0; i < np; ++i) {
    uint64_t s = ix->c2s[cd[i].idx];
    int seen = 0;
    for (uint64_t j = 0; j < ns; ++j) if (shard_order[j] == s) { seen = 1; break; }
    if (!seen) shard_order[ns++] = s;
  }
  size_t total = 0;
  for (uint64_t i = 0; i < np; ++i)
    if (ix->list_ok[cd[i].idx]) total += ix->list_len[cd[i].idx];
  cand *cs = (cand *)malloc(sizeof(cand) * (total + 1));
  size_t nc = 0;
  for (uint64_t si = 0; si < ns && rc == ORC_OK; ++si)
    for (uint64_t i = 0; i < np && rc == ORC_OK; ++i) { /* probe-rank order within shard */
      uint64_t c = cd[i].idx;
      if (ix->c2s[c] != shard_order[si] || !ix->list_ok[c]) continue;
      for (uint64_t v = 0; v < ix->list_len[c]; ++v) {
        float dist = orc_l2sq_scalar(q, ix->list_vec[c] + v * d, d);
        if (dist != dist) { rc = ORC_PANIC; break; }
        cs[nc].dist = dist; cs[nc].order = (uint32_t)nc; cs[nc].list = c; cs[nc].pos = v;
        nc++;
      }
    }
  if (rc == ORC_OK) {
    qsort(cs, nc, sizeof(cand), cmp_cand); /* stable sort :265 */
    uint64_t m = k < nc ? k : nc;
    for (uint64_t i = 0; i < m; ++i) {
      ids_out[i] = ix->list_meta[cs[i].list][3 * cs[i].pos + 1]; /* external_id :258 */
      dist_out[i] = cs[i].dist;
      if (vecs_out) memcpy(vecs_out + i * d, ix->list_vec[cs[i].list] + cs[i].pos * d, sizeof(float) * d);
    }
    *count = m;
  }
  free(cd); free(shard_order); free(cs);
  return rc;
}

/* Multi-GPU protocol checker (the placement is an extension: the reference has no device notion): the search
 * of rank `rank` of `world`, which keeps block b (64 vectors) of every list iff b % world == rank, with the
 * GLOBAL candidate-order key of every hit:
 * tie = (rank of the probe in the reference candidate order << 32) | position in list.
 * Merging the per-rank outputs by (dist, tie) must reproduce orc_index_search. */
int orc_index_search_partial(const orc_index *ix, const float *q, uint64_t k, uint64_t n_probe,
                             uint32_t rank, uint32_t world, uint64_t *ids_out, float *dist_out,
                             uint64_t *tie_out, uint64_t *count) {
  if (k == 0 || n_probe == 0) return ORC_INVALID_INPUT;
  size_t d = ix->dim;
  dist_idx *cd = (dist_idx *)malloc(sizeof(dist_idx) * (ix->k + 1));
  int rc = probe_lists(ix, q, n_probe, cd);
  if (rc != ORC_OK) { free(cd); return rc; }
  uint64_t np = n_probe < ix->k ? n_probe : ix->k;
  uint64_t *shard_order = (uint64_t *)malloc(sizeof(uint64_t) * (np + 1));
  uint64_t ns = 0;
  for (uint64_t i = 0; i < np; ++i) {
    uint64_t s = ix->c2s[cd[i].idx];
    int seen = 0;
    for (uint64_t j = 0; j < ns; ++j) if (shard_order[j] == s) { seen = 1; break; }
    if (!seen) shard_order[ns++] = s;
  }
  size_t total = 0;
  for (uint64_t i = 0; i < np; ++i) total += ix->list_len[cd[i].idx];
  cand *cs = (cand *)malloc(sizeof(cand) * (total + 1));
  uint64_t *ties = (uint64_t *)malloc(sizeof(uint64_t) * (total + 1));
  size_t nc = 0;
  uint64_t g = 0;
  for (uint64_t si = 0; si < ns; ++si)
    for (uint64_t i = 0; i < np; ++i) {
      uint64_t c = cd[i].idx;
      if (ix->c2s[c] != shard_order[si]) continue;
      uint64_t my_g = g++;
      if (!ix->list_ok[c]) continue;
      for (uint64_t v = 0; v < ix->list_len[c]; ++v) {
        if (world > 1 && ((v / 64) % world) != rank) continue;
        cs[nc].dist = orc_l2sq_scalar(q, ix->list_vec[c] + v * d, d);
        cs[nc].order = (uint32_t)nc; cs[nc].list = c; cs[nc].pos = v;
        ties[nc] = (my_g << 32) | v;
        nc++;
      }
    }
  qsort(cs, nc, sizeof(cand), cmp_cand);
  uint64_t m = k < nc ? k : nc;
  for (uint64_t i = 0; i < m; ++i) {
    ids_out[i] = ix->list_meta[cs[i].list][3 * cs[i].pos + 1];
    dist_out[i] = cs[i].dist;
    tie_out[i] = ties[cs[i].order];
  }
  *count = m;
  free(cd); free(shard_order); free(cs); free(ties);
  return ORC_OK;
}

int orc_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* bindings/python/src/lib.rs:74-97,179-202 */
int orc_index_search_batch(const orc_index *ix, const float *Q, uint64_t nq, uint64_t k,
                           uint64_t n_probe, int threads, float *D, int64_t *I) {
  if (k == 0 || n_probe == 0) return ORC_INVALID_INPUT;
  int rc_all = ORC_OK;
#ifdef _OPENMP
  if (threads <= 0) threads = omp_get_max_threads();
#else
  threads = 1;
#endif
#pragma omp parallel num_threads(threads)
  {
    uint64_t *ids = (uint64_t *)malloc(sizeof(uint64_t) * k);
    float *ds = (float *)malloc(sizeof(float) * k);
#pragma omp for schedule(dynamic, 4)
    for (long qi = 0; qi < (long)nq; ++qi) {
      uint64_t cnt = 0;
      int rc = orc_index_search(ix, Q + (size_t)qi * ix->dim, k, n_probe, ids, ds, NULL, &cnt);
      if (rc != ORC_OK) {
#pragma omp critical
        rc_all = rc;
        cnt = 0;
      }
      for (uint64_t j = 0; j < k; ++j) {
        D[(size_t)qi * k + j] = j < cnt ? ds[j] : INFINITY;
        I[(size_t)qi * k + j] = j < cnt ? (int64_t)ids[j] : -1;
      }
    }
    free(ids); free(ds);
  }
  return rc_all;
}
