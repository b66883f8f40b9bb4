// api.cpp — host-side mirror of the reference's public surface (src/api.rs) and the
// extern "C" boundary declared in include/vi_amd.h.
#include <sys/time.h>

#include <cmath>
#include <cstring>
#include <memory>
#include <new>
#include <stdexcept>
#include <string>

#include "device_index.hpp"
#include "kmeans.hpp"
#include "rng.hpp"
#include "shards.hpp"

namespace vi {

std::string &last_error_ref() {
  static thread_local std::string msg;
  return msg;
}

// mirrors VectorIndexer{cfg, index} — src/api.rs:96-99
struct Indexer {
  vi_config cfg{};
  std::string index_dir, shards_dir;
  IndexMeta meta;                       // IvfIndex{centroids, centroids_to_shard, dimension}
  std::unique_ptr<DeviceIndex> dev;     // HBM-resident lists (null until load/build)
  vi_build_stats build_stats{};         // phases of the last build on this handle
};

// No C++ exception may unwind through the C ABI into a Rust / ctypes caller (std::vector growth, std::string, a
// corrupt length field...): every extern "C" entry point that returns a status runs its body in here.
template <typename F>
static vi_status guarded(F &&body) noexcept {
  try {
    return body();
  } catch (const std::bad_alloc &) {
    return fail(VI_ERR_OTHER, "out of host memory");
  } catch (const std::exception &e) {
    return fail(VI_ERR_OTHER, "internal error: %s", e.what());
  } catch (...) {
    return fail(VI_ERR_OTHER, "internal error");
  }
}

static uint64_t unix_timestamp_secs() {  // src/utils.rs:109-114
  struct timeval tv;
  gettimeofday(&tv, nullptr);
  return (uint64_t)tv.tv_sec;
}

static vi_status attach_device(Indexer *ix) {
  auto dev = std::make_unique<DeviceIndex>();
  VI_TRY(device_index_load(ix->meta, ix->shards_dir, ix->cfg.device, ix->cfg.rank, ix->cfg.world_size, ix->cfg.placement,
                           dev.get()));
  ix->dev = std::move(dev);
  return VI_OK;
}

static double now_ms() {
  struct timeval tv;
  gettimeofday(&tv, nullptr);
  return tv.tv_sec * 1e3 + tv.tv_usec * 1e-3;
}

// IvfIndex::fit_with_paths (src/ivf_index.rs:58-177) + save_to (:274-294).  The points go to the GPU once and stay:
// k-means, the grouping of ids by list, the blocks of the resident index and the records of the shard files all come
// from that one device copy (list_build.hip); only the shard images travel back, to be written.
static vi_status fit_and_save(Indexer *ix, const float *X, const uint64_t *ext_ids, const uint64_t *timestamps,
                              uint64_t n) {
  const uint32_t dim = ix->cfg.dimension;
  const uint64_t seed = ix->cfg.seed ? ix->cfg.seed : 42;  // api.rs:143
  const uint64_t k = ix->cfg.nlist_override ? ix->cfg.nlist_override : vi_calculate_num_clusters(n);
  const uint64_t max_iters = vi_calculate_max_iterations(n);
  const uint64_t now = ix->cfg.now_secs ? ix->cfg.now_secs : unix_timestamp_secs();
  vi_build_stats &bs = ix->build_stats;
  bs = vi_build_stats{};
  bs.n = n; bs.nlist = k;
  const double t0 = now_ms();
  if (n > 0xFFFFFFFEull) return fail(VI_ERR_INVALID_INPUT, "more than 2^32 - 2 vectors");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(VI_ERR_DEVICE, "no HIP device visible: libvi_amd never falls back to the CPU");
  if (ix->cfg.device < 0 || ix->cfg.device >= ndev) return fail(VI_ERR_DEVICE, "device %d out of range (%d visible)", ix->cfg.device, ndev);
  VI_HIP(hipSetDevice(ix->cfg.device));
  hipStream_t st = nullptr;
  VI_HIP(hipStreamCreateWithFlags(&st, hipStreamDefault));
  struct StreamGuard { hipStream_t s; ~StreamGuard() { (void)hipStreamDestroy(s); } } guard{st};
  DevBuf<float> Xd, Cd;
  DevBuf<uint32_t> lab, order;
  DevBuf<uint64_t> ext_dev, ts_dev;
  VI_TRY(Xd.reserve(n * dim));
  VI_HIP(hipMemcpyAsync(Xd.p, X, n * dim * sizeof(float), hipMemcpyHostToDevice, st));
  if (ext_ids) { VI_TRY(ext_dev.reserve(n)); VI_HIP(hipMemcpyAsync(ext_dev.p, ext_ids, n * 8, hipMemcpyHostToDevice, st)); }
  if (timestamps) { VI_TRY(ts_dev.reserve(n)); VI_HIP(hipMemcpyAsync(ts_dev.p, timestamps, n * 8, hipMemcpyHostToDevice, st)); }
  VI_HIP(hipStreamSynchronize(st));
  const double t1 = now_ms();
  bs.ms_upload = (float)(t1 - t0);
  VI_TRY(Cd.reserve(k * dim));
  VI_TRY(lab.reserve(n));
  if (kmeans_mini_batch_device(ix->cfg.device, Xd.p, n, dim, k, max_iters, -1.0f, seed, (vi_assign_mode)ix->cfg.assign_mode,
                               Cd.p, lab.p, nullptr) != VI_OK)
    return fail(VI_ERR_PANIC, "Failed to run KMeans: %s", last_error_ref().c_str());  // .expect (ivf_index.rs:71)
  std::vector<float> C(k * dim);
  VI_HIP(hipMemcpy(C.data(), Cd.p, k * dim * sizeof(float), hipMemcpyDeviceToHost));
  const double t2 = now_ms();
  bs.ms_kmeans = (float)(t2 - t1);
  // IVF lists in ascending internal id (:94-101): ids grouped by label on the device
  std::vector<uint64_t> off;
  DevBuf<uint32_t> seg_dev;
  VI_TRY(group_ids_by_label_device(lab.p, n, k, order, seg_dev, &off, st));
  const double t3 = now_ms();
  bs.ms_group = (float)(t3 - t2);
  // super-centroids => shard of every list (:104-109)
  const uint64_t num_shards = (uint64_t)std::ceil(std::sqrt((float)k));
  const uint64_t super_seed = seed * 31ULL + 7ULL;
  std::vector<float> SC(num_shards * dim);
  std::vector<uint64_t> slab(k);
  KMeansOptions opt;
  opt.device = ix->cfg.device;
  opt.mode = (vi_assign_mode)ix->cfg.assign_mode;
  if (kmeans_mini_batch(C.data(), k, dim, num_shards, 100, -1.0f, super_seed, opt, SC.data(), slab.data(), nullptr) !=
      VI_OK)
    return fail(VI_ERR_PANIC, "Failed to run kmeans: %s", last_error_ref().c_str());
  const double t4 = now_ms();
  bs.ms_super = (float)(t4 - t3);
  // drop empty lists and renumber (:123-164)
  std::vector<uint64_t> newid(k, ~0ull);
  uint64_t kk = 0;
  for (uint64_t c = 0; c < k; ++c)
    if (off[c + 1] > off[c]) newid[c] = kk++;
  ix->meta.dimension = dim;
  ix->meta.centroids.assign(kk * dim, 0.0f);
  ix->meta.c2s.assign(kk, 0);
  std::vector<uint64_t> src_off(kk);
  std::vector<uint32_t> len(kk), lshard(kk);
  std::vector<std::vector<uint32_t>> of_shard(num_shards);  // kept lists of every shard, ascending
  for (uint64_t c = 0; c < k; ++c)
    if (newid[c] != ~0ull) {
      const uint64_t l = newid[c];
      std::memcpy(&ix->meta.centroids[l * dim], &C[c * dim], dim * sizeof(float));
      ix->meta.c2s[l] = slab[c];
      src_off[l] = off[c];
      len[l] = (uint32_t)(off[c + 1] - off[c]);
      lshard[l] = (uint32_t)slab[c];
      of_shard[slab[c]].push_back((uint32_t)l);
    }
  // every shard is written, also the ones that received no list (:118-120,166-171)
  {
    ShardExportWs ws;
    std::vector<uint64_t> cids, soff;
    std::vector<uint32_t> slen;
    std::vector<float> cvec;
    for (uint64_t s = 0; s < num_shards; ++s) {
      cids.clear(); soff.clear(); slen.clear(); cvec.clear();
      for (uint32_t l : of_shard[s]) {
        cids.push_back(l);
        soff.push_back(src_off[l]);
        slen.push_back(len[l]);
        cvec.insert(cvec.end(), &ix->meta.centroids[(uint64_t)l * dim], &ix->meta.centroids[(uint64_t)l * dim] + dim);
        bs.shard_bytes += (uint64_t)len[l] * record_stride(dim);
      }
      // a failed shard write is only reported, never fatal (ivf_index.rs:168-170)
      if (shard_export_device(ix->shards_dir, s, dim, cids, cvec.data(), soff, slen, Xd.p, order.p,
                              ext_ids ? ext_dev.p : nullptr, timestamps ? ts_dev.p : nullptr, now, ws, st) != VI_OK)
        fprintf(stderr, "Failed to write shard %llu to disk: %s\n", (unsigned long long)s, last_error_ref().c_str());
    }
  }
  VI_TRY(index_meta_save(ix->meta, ix->index_dir));
  const double t5 = now_ms();
  bs.ms_export = (float)(t5 - t4);
  // the resident index: straight from the device copy (a rank of a multi-GPU run keeps only its stripes: from the files)
  vi_status rc;
  if (ix->cfg.world_size > 1) {
    rc = attach_device(ix);
  } else {
    auto dev = std::make_unique<DeviceIndex>();
    rc = device_index_from_order(ix->cfg.device, dim, ix->meta.centroids.data(), kk, Xd.p, order.p, src_off, len, lshard,
                                 ext_ids ? ext_dev.p : nullptr, dev.get());
    if (rc == VI_OK) ix->dev = std::move(dev);
  }
  const double t6 = now_ms();
  bs.ms_index = (float)(t6 - t5);
  bs.ms_total = (float)(t6 - t0);
  bs.lists = kk;
  bs.shards = num_shards;
  return rc;
}

}  // namespace vi

using vi::fail;
using vi::Indexer;

struct vi_indexer {
  Indexer impl;
};

extern "C" {

const char *vi_last_error(void) { return vi::last_error_ref().c_str(); }
uint32_t vi_abi_version(void) { return VI_AMD_ABI_VERSION; }
int vi_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

// src/utils.rs:9-16
uint64_t vi_calculate_num_clusters(uint64_t n) {
  if (n < 10000) return (uint64_t)std::sqrt((double)n);
  if (n < 100000) return 2 * (uint64_t)std::ceil(std::sqrt((double)n));
  return 4 * (uint64_t)std::ceil(std::sqrt((double)n));
}
// src/utils.rs:18-26
uint64_t vi_calculate_max_iterations(uint64_t n) { return n < 10000 ? 300 : n < 100000 ? 100 : n < 1000000 ? 50 : 20; }
// src/kmeans.rs:83
uint64_t vi_minibatch_size(uint64_t n) {
  const uint64_t s = (uint64_t)std::sqrt((float)n);
  return s < 10 ? 10 : s > 256 ? 256 : s;
}

vi_status vi_l2sq_pairs(const float *a, const float *b, uint64_t n, uint32_t d, vi_sum_order order, float *out) {
  return vi::guarded([&]() -> vi_status {
  if ((n && (!a || !b || !out))) return fail(VI_ERR_INVALID_INPUT, "null pointer");
  return vi::l2sq_pairs_device(a, b, n, d, (int)order, out);
  });
}

vi_status vi_assign(const float *X, uint64_t n, uint32_t d, const float *C, uint64_t k, uint64_t seed,
                    vi_assign_mode mode, uint64_t *labels, float *dist_out) {
  return vi::guarded([&]() -> vi_status {
  if (n == 0) return VI_OK;
  if (!X || !C || !labels || d == 0 || k == 0) return fail(VI_ERR_INVALID_INPUT, "bad arguments to vi_assign");
  vi::KMeansOptions opt;
  opt.mode = mode;
  return vi::assign_points(X, n, d, C, k, seed, opt, labels, dist_out);
  });
}

vi_status vi_assign_device(int32_t device, const float *X_dev, uint64_t n, uint32_t d, const float *C_dev, uint64_t k,
                           uint64_t seed, vi_assign_mode mode, uint32_t *labels_dev, vi_assign_stats *stats) {
  return vi::guarded([&]() -> vi_status {
  if (n == 0) return VI_OK;
  if (!X_dev || !C_dev || !labels_dev || d == 0 || k == 0) return fail(VI_ERR_INVALID_INPUT, "bad arguments to vi_assign_device");
  return vi::assign_points_device(device, X_dev, n, d, C_dev, k, seed, mode, labels_dev, stats);
  });
}

vi_status vi_kmeans_mini_batch(const float *X, uint64_t n, uint32_t d, uint64_t k, uint64_t max_iters, float thr,
                               uint64_t seed, vi_assign_mode mode, float *C, uint64_t *labels, uint64_t *iters) {
  return vi::guarded([&]() -> vi_status {
  vi::KMeansOptions opt;
  opt.mode = mode;
  return vi::kmeans_mini_batch(X, n, d, k, max_iters, thr, seed, opt, C, labels, iters);
  });
}

vi_status vi_kmeans_parallel(const float *X, uint64_t n, uint32_t d, uint64_t k, uint64_t max_iters, float thr,
                             uint64_t seed, vi_assign_mode mode, float *C, uint64_t *labels, uint64_t *iters) {
  return vi::guarded([&]() -> vi_status {
  vi::KMeansOptions opt;
  opt.mode = mode;
  return vi::kmeans_parallel(X, n, d, k, max_iters, thr, seed, opt, C, labels, iters);
  });
}

vi_status vi_kmeans_mini_batch_device(int32_t device, const float *X_dev, uint64_t n, uint32_t d, uint64_t k,
                                      uint64_t max_iters, float thr, uint64_t seed, vi_assign_mode mode, float *C_dev,
                                      uint32_t *labels_dev, uint64_t *iters) {
  return vi::guarded([&]() -> vi_status {
  return vi::kmeans_mini_batch_device(device, X_dev, n, d, k, max_iters, thr, seed, mode, C_dev, labels_dev, iters);
  });
}

vi_status vi_kmeans_parallel_device(int32_t device, const float *X_dev, uint64_t n, uint32_t d, uint64_t k,
                                    uint64_t max_iters, float thr, uint64_t seed, vi_assign_mode mode, float *C_dev,
                                    uint32_t *labels_dev, uint64_t *iters) {
  return vi::guarded([&]() -> vi_status {
  return vi::kmeans_parallel_device(device, X_dev, n, d, k, max_iters, thr, seed, mode, C_dev, labels_dev, iters);
  });
}

vi_status vi_kmeans_mini_batch_train(int32_t device, const vi_row_source *rows, uint64_t n, uint32_t d, uint64_t k,
                                     uint64_t max_iters, float thr, uint64_t seed, float *C_dev, uint64_t *iters) {
  return vi::guarded([&]() -> vi_status {
  if (!rows) return fail(VI_ERR_INVALID_INPUT, "null row source");
  return vi::kmeans_mini_batch_train(device, *rows, n, d, k, max_iters, thr, seed, C_dev, iters);
  });
}

vi_status vi_kmeans_pp_init(int32_t device, const vi_row_source *rows, uint64_t n, uint32_t d, uint64_t k, uint64_t seed,
                            float *C_dev) {
  return vi::guarded([&]() -> vi_status {
  if (!rows) return fail(VI_ERR_INVALID_INPUT, "null row source");
  return vi::kmeans_pp_init_rows_entry(device, *rows, n, d, k, seed, C_dev);
  });
}

vi_status vi_kmeans_partial_sums_device(int32_t device, const float *X_dev, uint64_t n, uint32_t d,
                                        const uint32_t *labels_dev, uint64_t k, float *sums_dev, uint32_t *counts_dev) {
  return vi::guarded([&]() -> vi_status {
  return vi::kmeans_partial_sums_device(device, X_dev, n, d, labels_dev, k, sums_dev, counts_dev);
  });
}

vi_status vi_kmeans_finish_update_device(int32_t device, const float *sums_dev, const uint32_t *counts_dev, uint64_t k,
                                         uint32_t d, const float *C_prev_dev, float *C_new_dev, float *delta_out,
                                         uint32_t *empty_out, uint64_t *n_empty) {
  return vi::guarded([&]() -> vi_status {
  return vi::kmeans_finish_update_device(device, sums_dev, counts_dev, k, d, C_prev_dev, C_new_dev, delta_out, empty_out,
                                         n_empty);
  });
}

vi_status vi_kmeans_centroid_delta_device(int32_t device, const float *C_new_dev, const float *C_prev_dev, uint64_t k,
                                          uint32_t d, float *delta_out) {
  return vi::guarded([&]() -> vi_status {
  return vi::kmeans_centroid_delta(device, C_new_dev, C_prev_dev, k, d, delta_out);
  });
}

vi_rng *vi_rng_seed_from_u64(uint64_t seed) { return reinterpret_cast<vi_rng *>(new (std::nothrow) vi::StdRng(seed)); }
uint64_t vi_rng_gen_range(vi_rng *rng, uint64_t low, uint64_t high) {
  if (!rng || high <= low) return low;
  return reinterpret_cast<vi::StdRng *>(rng)->gen_range(low, high);
}
void vi_rng_free(vi_rng *rng) { delete reinterpret_cast<vi::StdRng *>(rng); }

vi_status vi_shard_save_to(const char *shards_dir, uint64_t shard_id, uint32_t dim, uint32_t num_lists,
                           const uint64_t *centroid_ids, const float *centroid_vecs, const uint64_t *list_off,
                           const uint64_t *ids, const uint64_t *ext_ids, const uint64_t *timestamps,
                           const float *vecs) {
  return vi::guarded([&]() -> vi_status {
  if (!shards_dir || !list_off) return fail(VI_ERR_INVALID_INPUT, "null pointer");
  return vi::shard_save_to(shards_dir, shard_id, dim, num_lists, centroid_ids, centroid_vecs, list_off, ids, ext_ids,
                           timestamps, vecs);
  });
}

vi_status vi_shard_get_centroid_vectors_from(const char *shards_dir, uint64_t shard_id, const uint64_t *centroid_ids,
                                             uint64_t n_req, uint32_t *dim_out, uint64_t *counts, float *centroid_out,
                                             uint64_t *metas_out, float *vecs_out) {
  return vi::guarded([&]() -> vi_status {
  if (!shards_dir) return fail(VI_ERR_INVALID_INPUT, "null pointer");
  vi::ShardFile f;
  VI_TRY(f.open(shards_dir, shard_id));
  const uint32_t dim = f.dim();
  if (dim_out) *dim_out = dim;
  const uint64_t stride = vi::record_stride(dim);
  uint64_t vbase = 0;
  for (uint64_t r = 0; r < n_req; ++r) {
    const vi::ShardListView *lv = f.find(centroid_ids[r]);
    if (!lv) return fail(VI_ERR_NOT_FOUND, "Centroid %llu not found", (unsigned long long)centroid_ids[r]);
    if (counts) counts[r] = lv->num_vectors;
    if (centroid_out) std::memcpy(centroid_out + r * dim, lv->centroid, 4ull * dim);
    for (uint32_t v = 0; v < lv->num_vectors; ++v) {
      const uint8_t *rec = lv->records + (uint64_t)v * stride;
      if (metas_out) std::memcpy(metas_out + (vbase + v) * 3, rec, vi::kVectorMetaBytes);
      if (vecs_out) std::memcpy(vecs_out + (vbase + v) * dim, rec + vi::kVectorMetaBytes, 4ull * dim);
    }
    vbase += lv->num_vectors;
  }
  return VI_OK;
  });
}

void vi_config_init(vi_config *cfg, uint32_t dimension) {
  std::memset(cfg, 0, sizeof(*cfg));
  cfg->dimension = dimension;
  cfg->default_k = 10;
  cfg->default_n_probe = 20;
  cfg->max_k = 10000;
  cfg->max_n_probe = 10000;
}

vi_status vi_indexer_new(const vi_config *cfg, vi_indexer **out) {
  return vi::guarded([&]() -> vi_status {
  if (!cfg || !out) return fail(VI_ERR_INVALID_INPUT, "null pointer");
  auto *h = new vi_indexer();
  h->impl.cfg = *cfg;
  h->impl.index_dir = cfg->index_dir ? cfg->index_dir : "index";
  h->impl.shards_dir = cfg->shards_dir ? cfg->shards_dir : "shards";
  h->impl.cfg.index_dir = nullptr;
  h->impl.cfg.shards_dir = nullptr;
  h->impl.meta.dimension = cfg->dimension;
  *out = h;
  return VI_OK;
  });
}

vi_status vi_indexer_load(const vi_config *cfg, vi_indexer **out) {
  return vi::guarded([&]() -> vi_status {
  vi_indexer *h = nullptr;
  VI_TRY(vi_indexer_new(cfg, &h));
  vi_status st = vi::index_meta_load(h->impl.index_dir, &h->impl.meta);
  if (st == VI_OK) st = vi::attach_device(&h->impl);
  if (st != VI_OK) { delete h; return st; }
  *out = h;
  return VI_OK;
  });
}

vi_status vi_indexer_build_from_records(vi_indexer *ix, const uint64_t *ext_ids, const float *values,
                                        const uint64_t *timestamps, const uint32_t *dims, uint64_t n) {
  return vi::guarded([&]() -> vi_status {
  if (!ix) return fail(VI_ERR_INVALID_INPUT, "null indexer");
  if (n == 0) return fail(VI_ERR_INVALID_INPUT, "no vectors provided");  // api.rs:116-118
  const uint32_t dim = ix->impl.cfg.dimension;
  if (dims)
    for (uint64_t i = 0; i < n; ++i)
      if (dims[i] != dim)  // api.rs:122-133
        return fail(VI_ERR_INVALID_INPUT, "vector dimension mismatch at index %llu: expected %u, got %u",
                    (unsigned long long)i, dim, dims[i]);
  if (!values) return fail(VI_ERR_INVALID_INPUT, "null values");
  return vi::fit_and_save(&ix->impl, values, ext_ids, timestamps, n);
  });
}

vi_status vi_indexer_build_from_vector_file(vi_indexer *ix, const char *vector_file) {
  return vi::guarded([&]() -> vi_status {
  if (!ix || !vector_file) return fail(VI_ERR_INVALID_INPUT, "invalid vector_file path");
  std::vector<vi::VectorFileRecord> recs;
  if (vi::read_vectors_from_file(vector_file, &recs) != VI_OK)
    return fail(VI_ERR_OTHER, "read_vectors_from_file: %s", vi::last_error_ref().c_str());  // api.rs:156
  if (recs.empty()) return fail(VI_ERR_INVALID_INPUT, "no vectors in vector_file");          // api.rs:158-163
  const uint32_t dim = ix->impl.cfg.dimension;
  for (size_t i = 0; i < recs.size(); ++i)
    if (recs[i].values.size() != dim)
      return fail(VI_ERR_INVALID_INPUT, "vector dimension mismatch at index %zu: expected %u, got %zu", i, dim,
                  recs[i].values.size());
  std::vector<float> X(recs.size() * (size_t)dim);
  std::vector<uint64_t> eid(recs.size()), ts(recs.size());
  for (size_t i = 0; i < recs.size(); ++i) {
    std::memcpy(&X[i * dim], recs[i].values.data(), dim * sizeof(float));
    eid[i] = recs[i].id;
    ts[i] = recs[i].meta;  // VectorStore::new(vectors): third field is the timestamp (api.rs:181)
  }
  return vi::fit_and_save(&ix->impl, X.data(), eid.data(), ts.data(), recs.size());
  });
}

static vi_status search_common(const vi_indexer *ix, uint64_t *k, uint64_t *n_probe) {
  if (!ix) return fail(VI_ERR_INVALID_INPUT, "null indexer");
  const vi_config &c = ix->impl.cfg;
  if (*k > c.max_k) *k = c.max_k;                    // api.rs:189
  if (*n_probe > c.max_n_probe) *n_probe = c.max_n_probe;  // api.rs:190
  return VI_OK;
}

vi_status vi_indexer_search(const vi_indexer *ix, const float *queries, uint64_t nq, uint32_t query_dim, uint64_t k,
                            uint64_t n_probe, float *D, int64_t *I, float *V, uint64_t *counts, uint64_t *k_out) {
  return vi::guarded([&]() -> vi_status {
  VI_TRY(search_common(ix, &k, &n_probe));
  if (k_out) *k_out = k;
  if (query_dim != ix->impl.cfg.dimension)  // api.rs:192-201
    return fail(VI_ERR_INVALID_INPUT, "query dimension mismatch: expected %u, got %u", ix->impl.cfg.dimension,
                query_dim);
  if (k == 0 || n_probe == 0)  // ivf_index.rs:197-202
    return fail(VI_ERR_INVALID_INPUT, "k and n_probe must be greater than 0");
  if (nq == 0) return VI_OK;
  if (!queries || !D || !I) return fail(VI_ERR_INVALID_INPUT, "null pointer");
  if (!ix->impl.dev) return fail(VI_ERR_DEVICE, "index is not resident on a GPU (build or load it first)");
  // NaN/Inf in a query makes partial_cmp().unwrap() panic in the reference (ivf_index.rs:215)
  for (uint64_t i = 0; i < nq * (uint64_t)query_dim; ++i)
    if (!std::isfinite(queries[i])) return fail(VI_ERR_PANIC, "non-finite query value at flat index %llu",
                                                (unsigned long long)i);
  vi::SearchIO io;
  io.queries = queries; io.nq = nq; io.k = k; io.n_probe = n_probe;
  io.D = D; io.I = I; io.V = V; io.counts = counts;
  return vi::device_index_search(*ix->impl.dev, io);
  });
}

vi_status vi_indexer_search_device(const vi_indexer *ix, const float *queries_dev, uint64_t nq, uint64_t k,
                                   uint64_t n_probe, float *D_dev, int64_t *I_dev, uint64_t *tie_dev) {
  return vi::guarded([&]() -> vi_status {
  VI_TRY(search_common(ix, &k, &n_probe));
  if (k == 0 || n_probe == 0) return fail(VI_ERR_INVALID_INPUT, "k and n_probe must be greater than 0");
  if (nq == 0) return VI_OK;
  if (!queries_dev || !D_dev || !I_dev) return fail(VI_ERR_INVALID_INPUT, "null pointer");
  if (!ix->impl.dev) return fail(VI_ERR_DEVICE, "index is not resident on a GPU (build or load it first)");
  vi::SearchIO io;
  io.queries = queries_dev; io.on_device = true; io.nq = nq; io.k = k; io.n_probe = n_probe;
  io.D = D_dev; io.I = I_dev; io.tie = tie_dev;
  return vi::device_index_search(*ix->impl.dev, io);
  });
}

vi_status vi_indexer_probe_device(const vi_indexer *ix, const float *queries_dev, uint64_t nq, uint64_t n_probe,
                                  uint32_t *probes_dev, uint32_t *order_dev, uint64_t *n_probe_eff) {
  return vi::guarded([&]() -> vi_status {
  uint64_t k = 1;
  VI_TRY(search_common(ix, &k, &n_probe));
  if (n_probe == 0) return fail(VI_ERR_INVALID_INPUT, "k and n_probe must be greater than 0");
  if (!ix->impl.dev) return fail(VI_ERR_DEVICE, "index is not resident on a GPU (build or load it first)");
  const uint64_t p_eff = std::min<uint64_t>(n_probe, ix->impl.meta.k());
  if (n_probe_eff) *n_probe_eff = p_eff;
  if (nq == 0 || p_eff == 0) return VI_OK;
  if (!queries_dev || !probes_dev || !order_dev) return fail(VI_ERR_INVALID_INPUT, "null pointer");
  vi::SearchIO io;
  io.queries = queries_dev; io.on_device = true; io.nq = nq; io.k = 1; io.n_probe = n_probe;
  io.probes_out = probes_dev; io.order_out = order_dev;
  return vi::device_index_search(*ix->impl.dev, io);
  });
}

vi_status vi_indexer_search_probed_device(const vi_indexer *ix, const float *queries_dev, uint64_t nq, uint64_t k,
                                          uint64_t n_probe_eff, const uint32_t *probes_dev,
                                          const uint32_t *order_dev, float *D_dev, int64_t *I_dev,
                                          uint64_t *tie_dev) {
  return vi::guarded([&]() -> vi_status {
  uint64_t n_probe = n_probe_eff;
  VI_TRY(search_common(ix, &k, &n_probe));
  if (k == 0 || n_probe == 0) return fail(VI_ERR_INVALID_INPUT, "k and n_probe must be greater than 0");
  if (nq == 0) return VI_OK;
  if (!queries_dev || !D_dev || !I_dev || !probes_dev || !order_dev) return fail(VI_ERR_INVALID_INPUT, "null pointer");
  if (!ix->impl.dev) return fail(VI_ERR_DEVICE, "index is not resident on a GPU (build or load it first)");
  if (n_probe_eff != std::min<uint64_t>(n_probe, ix->impl.meta.k()))
    return fail(VI_ERR_INVALID_INPUT, "n_probe_eff does not match this index (use the value vi_indexer_probe_device returned)");
  vi::SearchIO io;
  io.queries = queries_dev; io.on_device = true; io.nq = nq; io.k = k; io.n_probe = n_probe;
  io.D = D_dev; io.I = I_dev; io.tie = tie_dev;
  io.probes_in = probes_dev; io.order_in = order_dev;
  return vi::device_index_search(*ix->impl.dev, io);
  });
}

vi_status vi_merge_partials_device(int32_t device, uint64_t nq, uint64_t k, uint32_t parts, const float *D_parts,
                                   const int64_t *I_parts, const uint64_t *tie_parts, float *D_out, int64_t *I_out) {
  return vi::guarded([&]() -> vi_status {
  return vi::merge_partials_device(device, nq, k, parts, D_parts, I_parts, tie_parts, D_out, I_out);
  });
}

vi_status vi_merge_partials_packed_device(int32_t device, uint64_t nq, uint64_t k, uint32_t parts, const void *packed_dev,
                                          float *D_out, int64_t *I_out) {
  return vi::guarded([&]() -> vi_status {
  return vi::merge_partials_packed_device(device, nq, k, parts, packed_dev, D_out, I_out);
  });
}

uint64_t vi_packed_result_bytes(uint64_t nq, uint64_t k) { return (nq * k * 4 + 7) / 8 * 8 + 2 * nq * k * 8; }

uint32_t vi_indexer_dimension(const vi_indexer *ix) { return ix ? ix->impl.meta.dimension : 0; }
uint64_t vi_indexer_num_centroids(const vi_indexer *ix) { return ix ? ix->impl.meta.k() : 0; }
uint64_t vi_indexer_num_vectors(const vi_indexer *ix) { return ix && ix->impl.dev ? ix->impl.dev->nvec_resident : 0; }
uint64_t vi_indexer_num_shards(const vi_indexer *ix) { return ix && ix->impl.dev ? ix->impl.dev->nshards : 0; }

vi_status vi_indexer_centroids(const vi_indexer *ix, float *centroids_out, uint64_t *c2s_out) {
  return vi::guarded([&]() -> vi_status {
  if (!ix) return fail(VI_ERR_INVALID_INPUT, "null indexer");
  const vi::IndexMeta &m = ix->impl.meta;
  if (centroids_out && !m.centroids.empty()) std::memcpy(centroids_out, m.centroids.data(), m.centroids.size() * 4);
  if (c2s_out && !m.c2s.empty()) std::memcpy(c2s_out, m.c2s.data(), m.c2s.size() * 8);
  return VI_OK;
  });
}

void vi_indexer_free(vi_indexer *ix) { delete ix; }

vi_status vi_indexer_last_stats(const vi_indexer *ix, vi_search_stats *out) {
  return vi::guarded([&]() -> vi_status {
  if (!ix || !out || !ix->impl.dev) return fail(VI_ERR_INVALID_INPUT, "no stats");
  std::lock_guard<std::mutex> lock(ix->impl.dev->mu);
  *out = ix->impl.dev->last_stats;
  return VI_OK;
  });
}

vi_status vi_indexer_last_build_stats(const vi_indexer *ix, vi_build_stats *out) {
  return vi::guarded([&]() -> vi_status {
  if (!ix || !out) return fail(VI_ERR_INVALID_INPUT, "no stats");
  *out = ix->impl.build_stats;
  return VI_OK;
  });
}

void vi_indexer_enable_timing(vi_indexer *ix, int enable) {
  if (ix && ix->impl.dev) ix->impl.dev->timing = enable == 2 ? 2 : (enable != 0 ? 1 : 0);
}

}  // extern "C"
