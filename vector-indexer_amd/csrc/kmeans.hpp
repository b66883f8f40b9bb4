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

}  // namespace vi
