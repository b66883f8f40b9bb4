// kmeans.hpp — GPU k-means (assign / update on the device, RNG decisions on the host).
// Reference: src/kmeans.rs (run_kmeans_mini_batch :64-150, run_kmeans_parallel :15-60,
// assign_points_simd_parallel :445-459).
#pragma once
#include <cstdint>

#include "common.hpp"

namespace vi {

struct KMeansOptions {
  int device = 0;
  vi_assign_mode mode = VI_ASSIGN_REFERENCE;
};

vi_status assign_points(const float *X, uint64_t n, uint32_t d, const float *C, uint64_t k, uint64_t seed,
                        const KMeansOptions &opt, uint64_t *labels, float *dist_out);
vi_status assign_points_device(int device, const float *Xd, uint64_t n, uint32_t d, const float *Cd, uint64_t k,
                               uint64_t seed, vi_assign_mode mode, uint32_t *labels_dev, vi_assign_stats *stats);
vi_status kmeans_mini_batch(const float *X, uint64_t n, uint32_t d, uint64_t k, uint64_t max_iters, float thr,
                            uint64_t seed, const KMeansOptions &opt, float *C, uint64_t *labels, uint64_t *iters_run);
vi_status kmeans_parallel(const float *X, uint64_t n, uint32_t d, uint64_t k, uint64_t max_iters, float thr,
                          uint64_t seed, const KMeansOptions &opt, float *C, uint64_t *labels, uint64_t *iters_run);

vi_status kmeans_parallel_device(int device, const float *Xd, uint64_t n, uint32_t d, uint64_t k, uint64_t max_iters,
                                 float thr, uint64_t seed, vi_assign_mode mode, float *Cd, uint32_t *labels_dev,
                                 uint64_t *iters_run);
vi_status kmeans_mini_batch_device(int device, const float *Xd, uint64_t n, uint32_t d, uint64_t k, uint64_t max_iters,
                                   float thr, uint64_t seed, vi_assign_mode mode, float *Cd, uint32_t *labels_dev,
                                   uint64_t *iters_run);
vi_status kmeans_mini_batch_train(int device, const vi_row_source &rows, uint64_t n, uint32_t d, uint64_t k,
                                  uint64_t max_iters, float thr, uint64_t seed, float *Cd, uint64_t *iters_run);
vi_status kmeans_pp_init_rows_entry(int device, const vi_row_source &rows, uint64_t n, uint32_t d, uint64_t k,
                                    uint64_t seed, float *Cd);
vi_status kmeans_partial_sums_device(int device, const float *Xd, uint64_t n, uint32_t d, const uint32_t *labels_dev,
                                     uint64_t k, float *sums_dev, uint32_t *counts_dev);
vi_status kmeans_finish_update_device(int device, const float *sums_dev, const uint32_t *counts_dev, uint64_t k, uint32_t d,
                                      const float *C_prev_dev, float *C_new_dev, float *delta_out, uint32_t *empty_out,
                                      uint64_t *n_empty);
vi_status kmeans_centroid_delta(int device, const float *cur_dev, const float *prev_dev, uint64_t k, uint32_t d, float *delta);

}  // namespace vi
