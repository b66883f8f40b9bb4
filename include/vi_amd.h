/*
 * vi_amd.h — C ABI of libvi_amd.so, the MI355X-native (gfx950) implementation of the
 * distance hot path of NirajNair/vector-indexer.
 *
 * The reference is a pure-Rust crate with no FFI seam; the entry points below are the
 * calls its Rust host code (src/api.rs, src/ivf_index.rs, src/kmeans.rs) would bind
 * through `extern "C"` to move that path onto the GPU.  Each declaration cites the
 * reference interface it replaces (file:line relative to the reference repo).  The Rust
 * and Python bindings a maintainer would add are shown in INTEGRATION.md.
 *
 * Conventions
 *   - plain pointers + sizes, caller-allocated outputs, no exceptions/panics cross the ABI
 *   - every function returns a vi_status; vi_last_error() gives the thread-local message
 *   - vi_status mirrors the std::io::ErrorKind values the reference returns
 *   - all arrays are row-major contiguous, f32 / u64 / i64 little-endian host memory unless a
 *     parameter says "device"
 *   - entry points taking DEVICE pointers run on a library-owned stream created with
 *     hipStreamDefault, i.e. ordered after work already queued on the legacy null stream (where
 *     PyTorch-ROCm launches by default); callers using other streams must synchronise first.
 *     Every entry point returns only after its device work has completed.
 *   - a vi_indexer handle may be shared by several host threads: up to 4 searches run concurrently on one handle (each
 *     call holds its own stream and workspace; further callers wait), as the reference's &self search does
 *     (ivf_index_tests.rs:768-807)
 */
#ifndef VI_AMD_H
#define VI_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VI_AMD_ABI_VERSION 2

typedef enum vi_status {
  VI_OK = 0,
  VI_ERR_INVALID_INPUT = 1, /* io::ErrorKind::InvalidInput (api.rs:116-134,192-201; ivf_index.rs:197-202) */
  VI_ERR_NOT_FOUND = 2,     /* io::ErrorKind::NotFound (missing index.bin; shards.rs:257-265) */
  VI_ERR_INVALID_DATA = 3,  /* io::ErrorKind::InvalidData (shards.rs:215-231,310-316) */
  VI_ERR_OTHER = 4,         /* io::ErrorKind::Other (bincode wrap ivf_index.rs:280,312; shards.rs:193-213) */
  VI_ERR_IO = 5,            /* any other raw std::fs error */
  VI_ERR_PANIC = 6,         /* the reference would panic here (NaN in partial_cmp().unwrap(), ivf_index.rs:215,265) */
  VI_ERR_DEVICE = 7         /* no MI355X / HIP failure: the product never falls back to the CPU */
} vi_status;

/* Thread-local, NUL-terminated description of the last failure on this thread. */
const char *vi_last_error(void);
uint32_t vi_abi_version(void);
/* Number of visible HIP devices (0 on a CPU-only box; never an error). */
int vi_device_count(void);

/* ---- heuristics ------------------------------------------------------------------- */
/* replaces calculate_num_clusters — src/utils.rs:9-16 */
uint64_t vi_calculate_num_clusters(uint64_t num_vectors);
/* replaces calculate_max_iterations — src/utils.rs:18-26 */
uint64_t vi_calculate_max_iterations(uint64_t num_vectors);
/* mini-batch size rule — src/kmeans.rs:83 */
uint64_t vi_minibatch_size(uint64_t num_vectors);

/* ---- distance kernels ------------------------------------------------------------- */
typedef enum vi_sum_order {
  VI_ORDER_SCALAR = 0, /* euclidean_distance_squared — src/utils.rs:28-30 (search path) */
  VI_ORDER_LANES = 1   /* compute_distance_simd — src/kmeans.rs:377-419 (k-means path) */
} vi_sum_order;

/* out[i] = squared L2 between a[i,:] and b[i,:], n pairs of dimension d, computed on the
 * GPU with the reference's exact f32 summation order (bit-identical results). */
vi_status vi_l2sq_pairs(const float *a, const float *b, uint64_t n, uint32_t d, vi_sum_order order,
                        float *out);

/* ---- k-means ---------------------------------------------------------------------- */
typedef enum vi_assign_mode {
  /* reference behaviour: k <= 100 brute force, k > 100 hierarchical (approximate)
   * — assign_points_simd_parallel, src/kmeans.rs:445-459 */
  VI_ASSIGN_REFERENCE = 0,
  /* extension: exact nearest centroid for every k (MFMA -2·X·Cᵀ+‖c‖² filter + exact-order
   * re-check); equals assign_points_brute_force (src/kmeans.rs:462-470) bit for bit */
  VI_ASSIGN_EXACT = 1
} vi_assign_mode;

/* replaces assign_points_simd_parallel(&X,&C,&mut labels,seed) — src/kmeans.rs:445-450.
 * labels: n u64 (Rust usize).  dist_out (optional, n f32): lane-order distance to the label. */
vi_status vi_assign(const float *X, uint64_t n, uint32_t d, const float *C, uint64_t k, uint64_t seed,
                    vi_assign_mode mode, uint64_t *labels, float *dist_out);

/* Same assignment with DEVICE pointers (X_dev n x d, C_dev k x d, labels_dev n u32) on HIP device
 * `device`: nothing crosses PCIe.  stats (optional; zero-initialise it: it carries one optional input) reports the MFMA
 * filter's kernel time and how many rows needed the exact-order re-check. */
typedef struct vi_assign_stats {
  uint64_t n, k;
  uint64_t ambiguous_rows; /* rows re-evaluated exactly (VI_ASSIGN_EXACT / brute-force path) */
  uint32_t used_mfma;      /* 1 when the MFMA tiers ran (bf16 x 3, then f32 on the rows that one leaves ambiguous),
                              0 for the exact-order scan kernel alone */
  float ms_total;          /* wall time of the call incl. the exact re-check */
  float ms_filter;         /* HIP-event time of the first-tier MFMA kernel alone */
  uint64_t tier1_rows;     /* rows the first (bf16 x 3) tier left undecided; ambiguous_rows of them also failed the f32 tier */
  /* in (optional, zero = off): device buffer of ambiguous_cap u32 that receives the row indices counted in tier1_rows
   * (a superset of the rows re-evaluated exactly) — lets a test check exactly the rows the margins did not decide */
  uint32_t *ambiguous_rows_dev;
  uint64_t ambiguous_cap;
} vi_assign_stats;
vi_status vi_assign_device(int32_t device, const float *X_dev, uint64_t n, uint32_t d, const float *C_dev,
                           uint64_t k, uint64_t seed, vi_assign_mode mode, uint32_t *labels_dev,
                           vi_assign_stats *stats);

/* replaces run_kmeans_mini_batch(&data,k,max_iters,early_stop_threshold,seed)
 * — src/kmeans.rs:64-70.  early_stop_threshold < 0 means None (=> 1e-4).
 * centroids_out: k x d, labels_out: n.  iters_run optional.
 * Empty input => VI_ERR_INVALID_INPUT (kmeans.rs:72-77). */
vi_status vi_kmeans_mini_batch(const float *X, uint64_t n, uint32_t d, uint64_t k, uint64_t max_iters,
                               float early_stop_threshold, uint64_t seed, vi_assign_mode mode,
                               float *centroids_out, uint64_t *labels_out, uint64_t *iters_run);

/* replaces run_kmeans_parallel(...) — src/kmeans.rs:15-21 (same shape as above). */
vi_status vi_kmeans_parallel(const float *X, uint64_t n, uint32_t d, uint64_t k, uint64_t max_iters,
                             float early_stop_threshold, uint64_t seed, vi_assign_mode mode,
                             float *centroids_out, uint64_t *labels_out, uint64_t *iters_run);

/* The same two entry points with DEVICE pointers on HIP device `device` (X_dev n x d, centroids_dev k x d, labels_dev n
 * u32; labels_dev may be NULL for vi_kmeans_mini_batch_device to skip the final assignment): the points never cross PCIe
 * — config C3's 5 GB would take longer to upload than to assign. */
vi_status vi_kmeans_mini_batch_device(int32_t device, const float *X_dev, uint64_t n, uint32_t d, uint64_t k,
                                      uint64_t max_iters, float early_stop_threshold, uint64_t seed, vi_assign_mode mode,
                                      float *centroids_dev, uint32_t *labels_dev, uint64_t *iters_run);
vi_status vi_kmeans_parallel_device(int32_t device, const float *X_dev, uint64_t n, uint32_t d, uint64_t k,
                                    uint64_t max_iters, float early_stop_threshold, uint64_t seed, vi_assign_mode mode,
                                    float *centroids_dev, uint32_t *labels_dev, uint64_t *iters_run);

/* ---- k-means with the points SHARDED over the GPUs of a node (one process per GPU; extension: the reference has no
 *      distributed mode).  The collectives stay in the caller (RCCL through torch.distributed, as for search):
 *
 *   training   k-means++ seeding, the mini-batches and the empty-cluster re-seeds read only the rows the rand stream
 *              names (50 000 + k + iterations x (batch + empty clusters) of them).  vi_kmeans_mini_batch_train /
 *              vi_kmeans_pp_init run that loop over a vi_row_source: the caller's fetch_rows gathers the named rows
 *              from their owner ranks (a collective every rank enters with the same row list: every rank replays the
 *              same rand 0.8.5 stream; rank 0's centroids are broadcast afterwards, which pins them bit for bit).
 *   assignment vi_assign_device on every rank's own points (labels stay local) — the O(N k D) part.
 *   Lloyd      per iteration: vi_assign_device, vi_kmeans_partial_sums_device (this rank's sums k x d and counts k),
 *              all-reduce(sum) of both, vi_kmeans_finish_update_device (means, RMS movement, empty clusters);
 *              empty clusters are re-seeded with rows drawn from a vi_rng stream every rank replays (same seed, same
 *              draws) and fetched like the training rows.
 *              Sums combined over ranks associate differently from the reference's single pass: centroids agree
 *              to rounding (compare with a tolerance), labels of the first iteration exactly. */
typedef struct vi_row_source {
  void *ctx;
  /* out_dev[i, 0..d) = data row global_rows[i] for i < n_rows, complete on return; 0 = ok */
  int (*fetch_rows)(void *ctx, const uint64_t *global_rows, uint64_t n_rows, float *out_dev);
} vi_row_source;
/* the loop of run_kmeans_mini_batch (src/kmeans.rs:64-142) without its final assignment; centroids_dev k x d */
vi_status vi_kmeans_mini_batch_train(int32_t device, const vi_row_source *rows, uint64_t n_global, uint32_t d, uint64_t k,
                                     uint64_t max_iters, float early_stop_threshold, uint64_t seed,
                                     float *centroids_dev, uint64_t *iters_run);
/* kmeans_plus_plus_init (src/kmeans.rs:154-310) alone (the start of run_kmeans_parallel) */
vi_status vi_kmeans_pp_init(int32_t device, const vi_row_source *rows, uint64_t n_global, uint32_t d, uint64_t k,
                            uint64_t seed, float *centroids_dev);
/* update_centroids_parallel (src/kmeans.rs:674-719), this rank's share: sums_dev k x d f32, counts_dev k u32 */
vi_status vi_kmeans_partial_sums_device(int32_t device, const float *X_dev, uint64_t n_local, uint32_t d,
                                        const uint32_t *labels_dev, uint64_t k, float *sums_dev, uint32_t *counts_dev);
/* ... and what follows the all-reduce: C_new = sums / counts (zero rows for empty clusters), *delta_out =
 * compute_centroid_delta(C_new, C_prev) (src/kmeans.rs:334-351), empty_out (optional, host, k u32) / *n_empty = the
 * clusters handle_empty_clusters (src/kmeans.rs:313-331) re-seeds, ascending */
vi_status vi_kmeans_finish_update_device(int32_t device, const float *sums_dev, const uint32_t *counts_dev, uint64_t k,
                                         uint32_t d, const float *C_prev_dev, float *C_new_dev, float *delta_out,
                                         uint32_t *empty_out, uint64_t *n_empty);
/* compute_centroid_delta (src/kmeans.rs:334-351) of two device tables (after the re-seed, as the reference orders it) */
vi_status vi_kmeans_centroid_delta_device(int32_t device, const float *C_new_dev, const float *C_prev_dev, uint64_t k,
                                          uint32_t d, float *delta_out);
/* rand 0.8.5 StdRng::seed_from_u64(seed) and Rng::gen_range(low..high) on usize — the stream the reference draws
 * re-seed rows from (src/kmeans.rs:31,325) */
typedef struct vi_rng vi_rng;
vi_rng *vi_rng_seed_from_u64(uint64_t seed);
uint64_t vi_rng_gen_range(vi_rng *rng, uint64_t low, uint64_t high);
void vi_rng_free(vi_rng *rng);

/* ---- shard files ------------------------------------------------------------------ */
/* replaces Shard::save_to(&self, shards_dir) — src/shards.rs:68-177.  Lists flattened:
 * list i owns vectors [list_off[i], list_off[i+1]).  Byte-identical file layout. */
vi_status vi_shard_save_to(const char *shards_dir, uint64_t shard_id, uint32_t dim, uint32_t num_lists,
                           const uint64_t *centroid_ids, const float *centroid_vecs,
                           const uint64_t *list_off, const uint64_t *ids, const uint64_t *ext_ids,
                           const uint64_t *timestamps, const float *vecs);

/* replaces Shard::get_centroid_vectors_from(shards_dir, shard_id, &centroid_ids)
 * — src/shards.rs:188-349.  Call once with the output buffers NULL to get counts[i]
 * (vectors in requested list i) and *dim_out, then again with buffers:
 * centroid_out n_req x dim; metas_out 3 u64 {id, external_id, timestamp} per vector;
 * vecs_out dim f32 per vector, in request order. */
vi_status vi_shard_get_centroid_vectors_from(const char *shards_dir, uint64_t shard_id,
                                             const uint64_t *centroid_ids, uint64_t n_req,
                                             uint32_t *dim_out, uint64_t *counts, float *centroid_out,
                                             uint64_t *metas_out, float *vecs_out);

/* ---- VectorIndexer (src/api.rs) ---------------------------------------------------- */
/* mirrors VectorIndexerConfig — src/api.rs:8-54 — plus clearly marked extensions */
typedef struct vi_config {
  uint32_t dimension;
  const char *index_dir;    /* NULL => "index"  (api.rs:36) */
  const char *shards_dir;   /* NULL => "shards" (api.rs:37) */
  uint64_t default_k;       /* 10      (api.rs:38) */
  uint64_t default_n_probe; /* 20      (api.rs:39) */
  uint64_t max_k;           /* 10 000  (api.rs:40) */
  uint64_t max_n_probe;     /* 10 000  (api.rs:41) */
  /* --- extensions (all zero = reference behaviour) --- */
  uint64_t nlist_override;  /* 0 => calculate_num_clusters(N) (ivf_index.rs:59) */
  uint64_t seed;            /* 0 => 42 (api.rs:143,183) */
  int32_t assign_mode;      /* vi_assign_mode for the build's final assignment */
  int32_t device;           /* HIP device ordinal */
  /* multi-GPU partition: this process keeps a stripe of every inverted list resident (block b of 64 vectors
   * lives on rank b % world_size; coarse table replicated).  world_size 0/1 => everything. */
  int32_t rank;
  int32_t world_size;
  uint64_t now_secs;        /* 0 => wall clock; else value used for timestamp==0 records
                               (vector_store.rs:36-40) */
  /* multi-GPU partition rule: 0 = stripes (above; balances whatever the skew); 1 = shard placement: whole shard
   * files — the reference's super-centroid grouping of lists, src/ivf_index.rs:104-164 — are dealt to the ranks greedily
   * by bytes (largest first to the least loaded rank), a list is scanned by the one rank that holds its shard */
  int32_t placement;
  int32_t reserved0;
} vi_config;

/* VectorIndexerConfig::new(dimension) — src/api.rs:33-43 */
void vi_config_init(vi_config *cfg, uint32_t dimension);

typedef struct vi_indexer vi_indexer;

/* VectorIndexer::new(cfg) — src/api.rs:103-106 */
vi_status vi_indexer_new(const vi_config *cfg, vi_indexer **out);
/* VectorIndexer::load(cfg) — src/api.rs:109-112 (+ uploads the lists to HBM) */
vi_status vi_indexer_load(const vi_config *cfg, vi_indexer **out);
/* VectorIndexer::build_from_records(self, records) — src/api.rs:115-146.
 * values: n x dimension; ext_ids: n (NULL => 0..n-1); timestamps: n (NULL or 0 => now).
 * dims (optional, n): per-record length, to reproduce the "vector dimension mismatch at
 * index {i}" error (api.rs:122-133); NULL => all equal cfg.dimension.
 * Device memory: the build keeps the uploaded points (4 n D bytes, + 16 n for ids and timestamps when given) resident
 * until the searchable index built from them (another 4 n D for the blocks + 4 n D of bf16 images, + 1 n D for 8-bit data)
 * is complete: its peak is about twice what vi_indexer_load of the same index needs; the grouping sort adds 16 n bytes
 * and the largest shard's staging image transiently.  A data set that fills more than ~40 % of HBM should be built in
 * parts or loaded from shard files written elsewhere. */
vi_status vi_indexer_build_from_records(vi_indexer *ix, const uint64_t *ext_ids, const float *values,
                                        const uint64_t *timestamps, const uint32_t *dims, uint64_t n);
/* VectorIndexer::build_from_vector_file(self, path) — src/api.rs:149-186 */
vi_status vi_indexer_build_from_vector_file(vi_indexer *ix, const char *vector_file);

/* VectorIndexer::search(&self, SearchRequest) — src/api.rs:188-222 — batched over nq queries
 * (extension: the reference takes one query per call; bindings/python/src/lib.rs:74-97 loops).
 * k and n_probe are clamped to max_k / max_n_probe first (api.rs:189-190), then k==0 or
 * n_probe==0 => VI_ERR_INVALID_INPUT (ivf_index.rs:197-202); query_dim != dimension =>
 * VI_ERR_INVALID_INPUT (api.rs:192-201).
 * Outputs (row stride = clamped k, which is returned in *k_out if non-NULL):
 *   D  nq x k f32 squared-L2, padded +inf     (lib.rs:179-187)
 *   I  nq x k i64 external ids, padded -1
 *   V  optional nq x k x dimension f32 (include_vectors; api.rs:213-217), zero padded
 *   counts optional nq: results actually found per query */
vi_status vi_indexer_search(const vi_indexer *ix, const float *queries, uint64_t nq, uint32_t query_dim,
                            uint64_t k, uint64_t n_probe, float *D, int64_t *I, float *V,
                            uint64_t *counts, uint64_t *k_out);

/* Same search, but queries/D/I are DEVICE pointers on the indexer's GPU and nothing is copied
 * to the host; tie (optional, nq x k u64) receives the reference candidate-order key
 * ((shard-visit-order<<32)|position) needed to merge per-GPU partial results exactly.
 * Used by the multi-GPU path (all-gather of per-rank top-k over RCCL). */
vi_status vi_indexer_search_device(const vi_indexer *ix, const float *queries_dev, uint64_t nq,
                                   uint64_t k, uint64_t n_probe, float *D_dev, int64_t *I_dev,
                                   uint64_t *tie_dev);

/* Multi-GPU: the two halves of vi_indexer_search_device as separate calls, so that the coarse step — identical on
 * every rank, since the centroid table is replicated — can be SPLIT over the ranks by query instead of repeated:
 *   vi_indexer_probe_device          ivf_index.rs:205-220 for a slice of the batch: the n_probe_eff = min(n_probe,
 *                                    #centroids) nearest lists of each query in (distance, centroid index) order,
 *                                    and order[q][r] = rank of probe r in the reference's candidate order (shard
 *                                    visiting order, ivf_index.rs:223-262) — what the tie keys are built from;
 *   (all-gather of probes / order over the ranks)
 *   vi_indexer_search_probed_device  ivf_index.rs:223-274 for the whole batch against this rank's stripes with the
 *                                    given probe lists; outputs as vi_indexer_search_device.
 * Both return with their outputs complete; all pointers are device pointers.  Any k and n_probe the single-GPU entry
 * accepts (k > 128 or n_probe > 64 take the sort-everything path there and here). */
vi_status vi_indexer_probe_device(const vi_indexer *ix, const float *queries_dev, uint64_t nq, uint64_t n_probe,
                                  uint32_t *probes_dev, uint32_t *order_dev, uint64_t *n_probe_eff);
vi_status vi_indexer_search_probed_device(const vi_indexer *ix, const float *queries_dev, uint64_t nq, uint64_t k,
                                          uint64_t n_probe_eff, const uint32_t *probes_dev,
                                          const uint32_t *order_dev, float *D_dev, int64_t *I_dev,
                                          uint64_t *tie_dev);

/* Merge `parts` per-rank partial results (each nq x k, device pointers laid out
 * [part][nq][k]) into the global top-k with the reference's stable order. */
vi_status vi_merge_partials_device(int32_t device, uint64_t nq, uint64_t k, uint32_t parts,
                                   const float *D_parts, const int64_t *I_parts,
                                   const uint64_t *tie_parts, float *D_out, int64_t *I_out);

/* The same merge for results PACKED per rank as [D f32 nq*k | pad to 8 B | I i64 nq*k | tie u64 nq*k]
 * (vi_packed_result_bytes(nq, k) bytes; point vi_indexer_search*_device's D / I / tie outputs into one such buffer):
 * the ranks then exchange their partial results with ONE all-gather instead of three. */
uint64_t vi_packed_result_bytes(uint64_t nq, uint64_t k);
vi_status vi_merge_partials_packed_device(int32_t device, uint64_t nq, uint64_t k, uint32_t parts,
                                          const void *packed_dev, float *D_out, int64_t *I_out);

/* accessors */
uint32_t vi_indexer_dimension(const vi_indexer *ix);
uint64_t vi_indexer_num_centroids(const vi_indexer *ix);
uint64_t vi_indexer_num_vectors(const vi_indexer *ix); /* resident on this rank */
uint64_t vi_indexer_num_shards(const vi_indexer *ix);
/* centroids (k' x dim) and centroids_to_shard (k') of the loaded index */
vi_status vi_indexer_centroids(const vi_indexer *ix, float *centroids_out, uint64_t *c2s_out);
void vi_indexer_free(vi_indexer *ix);

/* ---- instrumentation (bench.py / profiling) ---------------------------------------- */
typedef struct vi_search_stats {
  uint64_t nq, k, n_probe_eff;
  uint64_t coarse_candidates;  /* nq * k' */
  uint64_t scanned_vectors;    /* Σ over queries of Σ len(probed lists resident here) */
  uint64_t scan_items;         /* (list, query-group) work items of the list-scan kernel */
  float ms_total, ms_coarse, ms_group, ms_scan, ms_merge; /* HIP-event times on the search stream */
  uint64_t fallback_queries;   /* always 0 (kept for ABI stability: the MFMA path has no fallback) */
  uint64_t filter_tile_blocks; /* MFMA path: (128-query group, 64-vector block) tiles ranked; 0 on the VALU engine */
  uint64_t filter_rechecked;   /* VI_FILTER_STATS=1: (query, vector) pairs re-evaluated in exact order */
  uint64_t filter_accepted;    /* VI_FILTER_STATS=1: block records consulted */
  uint64_t rank_mode;          /* list scan of the last search: 0 exact-order VALU engine, 1 f32 MFMA,
                                  2 bf16 x 3 MFMA, 3 bf16 MFMA on hi planes only (bf16-exact stored values),
                                  4 bf16 MFMA on the hi planes of real-valued lists (wider margin, more exact re-evaluations),
                                  5 / 6 = 2 / 4 with the images taken about the mean of the stored vectors (real-valued
                                  lists far from the origin: the margins scale with the spread, not the offset) */
  uint64_t group_queries;      /* MFMA path: queries per rank work item (128, or 32 when lists are probed by few) */
} vi_search_stats;
/* phases of the most recent build on this handle (wall-clock ms): the points are uploaded once; k-means, the grouping of
 * ids by list, the shard export and the resident index all work from that device copy */
typedef struct vi_build_stats {
  uint64_t n, nlist, lists, shards;  /* points, requested lists, non-empty lists, shard files */
  uint64_t shard_bytes;              /* record bytes written to the shard files */
  float ms_total, ms_upload, ms_kmeans, ms_group, ms_super, ms_export, ms_index;
} vi_build_stats;
vi_status vi_indexer_last_build_stats(const vi_indexer *ix, vi_build_stats *out);

/* stats of the most recent search on this handle (timing collected only if enabled) */
vi_status vi_indexer_last_stats(const vi_indexer *ix, vi_search_stats *out);
/* enable: 0 off; 1 HIP events at every phase boundary of a search (ms_total / ms_coarse / ms_group / ms_scan / ms_merge);
 * 2 around the list-rank kernel only (ms_scan; the others read 0) — every event record is a barrier packet on the search
 * stream, five of them cost about 0.01 ms per search */
void vi_indexer_enable_timing(vi_indexer *ix, int enable);

#ifdef __cplusplus
}
#endif
#endif /* VI_AMD_H */
