// assign_mfma.hpp — exact nearest-centroid assignment with an f32-MFMA filter (assign_mfma.hip).
#pragma once
#include <cstdint>

#include "common.hpp"

namespace vi {

struct MfmaAssignStats {
  uint64_t ambiguous_rows = 0;  // rows re-evaluated by the exact-order scan
  uint64_t tier1_rows = 0;      // rows the first MFMA pass left undecided
  uint64_t tier2_rows = 0;      // ... of which went through the f32 MFMA pass (the rest straight to the exact scan)
  float ms_filter = 0.0f;       // HIP-event time of the MFMA kernel launches
  uint32_t *export_rows = nullptr;  // optional (device): receives the first-tier ambiguous row indices
  uint64_t export_cap = 0, exported = 0;
};

struct MfmaAssignWs {
  DevBuf<float> cn;
  DevBuf<uint32_t> namb, amb_list;
  DevBuf<uint32_t> img;   // bf16 hi/lo images of the centroid tiles
  DevBuf<uint32_t> img_hi, cand, cand_cnt;  // hi-only candidate sweep: hi image, candidate lists, their lengths
  DevBuf<float> cand_thr;                   // ... and every point's final threshold
  DevBuf<float> cnpad;    // centroid norms padded to whole tiles (+inf)
  DevBuf<float> xc;       // second tier: gathered ambiguous rows, their labels, what stays ambiguous
  DevBuf<uint32_t> lab_c, amb_list2, namb2;
};

// Re-evaluates rows rows_dev[0..nrows) of X (row-major, device) exactly over all centroids and
// writes labels[rows_dev[i]].  Provided by the caller (kmeans.hip: exact-order scan kernel).
typedef vi_status (*ExactRowsFn)(void *ctx, const float *X, const uint32_t *rows_dev, uint32_t nrows,
                                 uint32_t *labels);

// d <= 128 and k >= 128: the shapes the register-resident X tile covers
bool mfma_assign_supported(uint64_t n, uint64_t k, uint32_t d);

// labels_dev[i] = arg-min_c compute_distance_simd(X_i, C_c) with strict '<' — bit-identical to
// assign_points_brute_force (src/kmeans.rs:462-470).  All pointers are device pointers.
vi_status mfma_assign_device(const float *Xd, uint64_t n, const float *Cd, uint64_t k, uint32_t d,
                             uint32_t *labels_dev, MfmaAssignWs &ws, hipStream_t st, ExactRowsFn exact, void *exact_ctx,
                             MfmaAssignStats *stats);

}  // namespace vi
