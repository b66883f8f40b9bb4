// kmeans.hip — placeholder while the GPU k-means path is being built.
#include "kmeans.hpp"
namespace vi {
vi_status assign_points(const float *, uint64_t, uint32_t, const float *, uint64_t, uint64_t, const KMeansOptions &,
                        uint64_t *, float *) { return fail(VI_ERR_OTHER, "k-means path not built yet"); }
vi_status kmeans_mini_batch(const float *, uint64_t, uint32_t, uint64_t, uint64_t, float, uint64_t,
                            const KMeansOptions &, float *, uint64_t *, uint64_t *) {
  return fail(VI_ERR_OTHER, "k-means path not built yet");
}
vi_status kmeans_parallel(const float *, uint64_t, uint32_t, uint64_t, uint64_t, float, uint64_t,
                          const KMeansOptions &, float *, uint64_t *, uint64_t *) {
  return fail(VI_ERR_OTHER, "k-means path not built yet");
}
}  // namespace vi
