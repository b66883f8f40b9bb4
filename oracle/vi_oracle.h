/*
 * vi_oracle.h — CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the distance hot path of NirajNair/vector-indexer
 * (reference mounted at /root/reference; citations are file:line into it).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The product library (libvi_amd.so) never links it.
 *
 * PARITY PIN STATUS
 *   - The reference is Rust and cannot be compiled here (no cargo/rustc), and
 *     its own tests hold no golden vectors for distances / k-means / search.
 *     What IS pinned: the shard byte layout (src/shards.rs:22-51,68-177; the
 *     round-trip scenarios of tests/shards_tests.rs), the heuristics tables
 *     (src/utils.rs:9-26) and the exhaustive-probe == brute-force family
 *     (tests/api_tests.rs:40-92).  tests/golden/ holds vectors for those,
 *     produced by tests/golden/make_golden.py with an independent numpy
 *     float32 implementation.
 *   - Third-party arithmetic that is NOT in /root/reference is restated from
 *     the crates' published algorithms and is "parity unpinned":
 *       rand 0.8.5 / rand_chacha 0.3.1 / rand_core 0.6.4  (StdRng = ChaCha12,
 *         seed_from_u64 = PCG32 expansion, gen_range = widening-multiply
 *         rejection, shuffle, choose_multiple, WeightedIndex<f32>)
 *       wide 0.7.33 (f32x8/f32x4 lane arithmetic; reduce_add order on the
 *         default SSE2 build taken as ((l0+l1)+l2)+l3 per f32x4 and
 *         f32x8 = reduce(lo4) + reduce(hi4))
 *       bincode 2.0.1 standard config + ndarray 0.15 serde (index/index.bin)
 */
#ifndef VI_ORACLE_H
#define VI_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* error codes mirror std::io::ErrorKind as used by the reference */
enum {
  ORC_OK = 0,
  ORC_INVALID_INPUT = 1, /* ErrorKind::InvalidInput */
  ORC_NOT_FOUND = 2,     /* ErrorKind::NotFound */
  ORC_INVALID_DATA = 3,  /* ErrorKind::InvalidData */
  ORC_OTHER = 4,         /* ErrorKind::Other */
  ORC_IO = 5,            /* raw fs error */
  ORC_PANIC = 6          /* the reference would panic (NaN in sort, ...) */
};

/* ---- heuristics: src/utils.rs:9-26, src/kmeans.rs:83,483, src/ivf_index.rs:104 */
uint64_t orc_calculate_num_clusters(uint64_t n);
uint64_t orc_calculate_max_iterations(uint64_t n);
uint64_t orc_minibatch_size(uint64_t n);
uint64_t orc_meta_k(uint64_t k);
uint64_t orc_num_shards(uint64_t k);

/* ---- distance kernels */
/* src/utils.rs:28-30: sequential left-to-right f32 sum of (x-y)^2 */
float orc_l2sq_scalar(const float *a, const float *b, size_t d);
/* src/kmeans.rs:377-419: 8-lane + 4-lane + tail accumulation */
float orc_l2sq_simd(const float *p, const float *c, size_t d);

/* ---- rand 0.8.5 StdRng restatement (exposed for tests) */
typedef struct {
  uint32_t key[8];
  uint64_t counter;
  uint32_t buf[64];
  uint32_t index;
} orc_rng;
void orc_rng_seed_from_u64(orc_rng *r, uint64_t seed);
void orc_rng_from_seed(orc_rng *r, const uint8_t seed[32]);
uint32_t orc_rng_next_u32(orc_rng *r);
uint64_t orc_rng_next_u64(orc_rng *r);
uint64_t orc_rng_gen_range_usize(orc_rng *r, uint64_t low, uint64_t high); /* [low,high) */
void orc_rng_shuffle_u64(orc_rng *r, uint64_t *v, uint64_t n);
/* raw ChaCha block with a configurable round count (20 => RFC 7539 vectors) */
void orc_chacha_block(const uint32_t key[8], uint64_t counter, uint64_t stream,
                      int rounds, uint32_t out[16]);

/* ---- k-means: src/kmeans.rs */
/* find_nearest_centroid :355-373 */
void orc_find_nearest_centroid(const float *p, const float *C, size_t k, size_t d,
                               uint64_t *best, float *best_dist);
/* assign_points_brute_force :462-470 */
void orc_assign_brute_force(const float *X, size_t n, size_t d, const float *C,
                            size_t k, uint64_t *labels);
/* assign_points_hierarchical :474-581 (+ :584-672) */
void orc_assign_hierarchical(const float *X, size_t n, size_t d, const float *C,
                             size_t k, uint64_t seed, uint64_t *labels);
/* assign_points_simd_parallel :445-459 (k > 100 => hierarchical) */
void orc_assign(const float *X, size_t n, size_t d, const float *C, size_t k,
                uint64_t seed, uint64_t *labels);
/* build_centroid_hierarchy :584-648 (exposed so the GPU path can be compared
 * stage by stage): meta (meta_k x d), c2m (k) */
void orc_build_centroid_hierarchy(const float *C, size_t k, size_t d, size_t meta_k,
                                  uint64_t seed, float *meta, uint64_t *c2m);
/* kmeans_plus_plus_init :154-310 */
void orc_kmeans_pp_init(const float *X, size_t n, size_t d, size_t k, uint64_t seed,
                        float *C);
/* update_centroids_parallel :674-719 */
void orc_update_centroids(const float *X, size_t n, size_t d, const uint64_t *labels,
                          size_t k, float *C_new, uint64_t *counts);
/* run_kmeans_mini_batch :64-150.  thr < 0 => None => 1e-4.
 * force_brute != 0 replaces the final assign by exact brute force (extension
 * used to check the GPU "exact" mode). iters_run (optional) = iterations done.
 * labels == NULL skips the final assignment (:146-147): the training loop alone */
int orc_kmeans_mini_batch(const float *X, size_t n, size_t d, size_t k,
                          size_t max_iters, float thr, uint64_t seed, int force_brute,
                          float *C, uint64_t *labels, uint64_t *iters_run);
/* run_kmeans_parallel :15-60 */
int orc_kmeans_parallel(const float *X, size_t n, size_t d, size_t k, size_t max_iters,
                        float thr, uint64_t seed, int force_brute, float *C,
                        uint64_t *labels, uint64_t *iters_run);

/* ---- shard files: src/shards.rs */
/* Shard::save_to :68-177.  Lists are given flattened: list i owns vectors
 * [list_off[i], list_off[i+1]) of (ids, ext_ids, timestamps, vecs). */
int orc_shard_save_to(const char *shards_dir, uint64_t shard_id, uint32_t dim,
                      uint32_t num_lists, const uint64_t *centroid_ids,
                      const float *centroid_vecs, const uint64_t *list_off,
                      const uint64_t *ids, const uint64_t *ext_ids,
                      const uint64_t *timestamps, const float *vecs);
/* Shard::get_centroid_vectors_from :188-349.  Two-call protocol: first with
 * out pointers NULL to obtain counts[i] (vectors of requested list i), then with
 * buffers.  centroid_out: n_req x dim. metas: 3 u64 per vector. */
int orc_shard_get_centroid_vectors_from(const char *shards_dir, uint64_t shard_id,
                                        const uint64_t *centroid_ids, size_t n_req,
                                        uint32_t *dim_out, uint64_t *counts,
                                        float *centroid_out, uint64_t *metas_out,
                                        float *vecs_out);

/* ---- IVF index: src/ivf_index.rs, src/api.rs */
typedef struct orc_index orc_index;
/* IvfIndex::fit_with_paths :58-177 + save_to :274-294.
 * nlist_override = 0 => calculate_num_clusters(n) (reference behaviour).
 * timestamps[i]==0 => "now" (vector_store.rs:36-40); pass a fixed now_secs for
 * reproducibility.  Writes shard files + index/index.bin and returns the index
 * with lists resident in RAM. */
int orc_index_build(const float *X, const uint64_t *ext_ids, const uint64_t *timestamps,
                    size_t n, uint32_t dim, uint64_t nlist_override, uint64_t seed,
                    uint64_t now_secs, int force_brute, const char *index_dir,
                    const char *shards_dir, orc_index **out);
/* load_index_from :302-316 (+ all shard lists preloaded into RAM) */
int orc_index_load(const char *index_dir, const char *shards_dir, orc_index **out);
void orc_index_free(orc_index *ix);
uint64_t orc_index_num_centroids(const orc_index *ix);
uint32_t orc_index_dimension(const orc_index *ix);
uint64_t orc_index_num_shards(const orc_index *ix);
void orc_index_centroids(const orc_index *ix, float *C_out, uint64_t *c2s_out);
uint64_t orc_index_list_len(const orc_index *ix, uint64_t list);
/* IvfIndex::search_with_paths :190-267 for one query.  Shard visiting order is
 * fixed to "first appearance in the probe list" (the reference iterates a
 * HashSet, i.e. any order is a possible outcome).
 * out arrays sized k; *count = results written. vecs_out optional (k x dim). */
int orc_index_search(const orc_index *ix, const float *q, uint64_t k, uint64_t n_probe,
                     uint64_t *ids_out, float *dist_out, float *vecs_out, uint64_t *count);
/* batch = loop over queries (bindings/python/src/lib.rs:74-97), OpenMP over
 * queries with `threads` threads (<=0 => omp default).  D padded with +inf and
 * I with -1 (lib.rs:179-187). */
int orc_index_search_batch(const orc_index *ix, const float *Q, uint64_t nq, uint64_t k,
                           uint64_t n_probe, int threads, float *D, int64_t *I);
/* coarse step only (ivf_index.rs:205-220): probe list for one query */
int orc_index_probe(const orc_index *ix, const float *q, uint64_t n_probe,
                    uint64_t *probes_out, uint64_t *count);
/* per-rank partial search with global tie keys (multi-GPU protocol checker) */
int orc_index_search_partial(const orc_index *ix, const float *q, uint64_t k, uint64_t n_probe,
                             uint32_t rank, uint32_t world, uint64_t *ids_out, float *dist_out,
                             uint64_t *tie_out, uint64_t *count);
int orc_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
