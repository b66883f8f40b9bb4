// kmeans.hip — GPU k-means of the IVF build path.
//
// Reference: src/kmeans.rs — run_kmeans_mini_batch (:64-150), run_kmeans_parallel (:15-60),
// kmeans_plus_plus_init (:154-310), assign_points_{brute_force,hierarchical} (:462-581),
// update_centroids_{parallel,mini_batch} (:674-787), handle_empty_clusters (:313-331),
// compute_centroid_delta (:334-351).
//
// Split of work:
//   GPU  every distance (exact lane-structured order of compute_distance_simd, kmeans.rs:377-419),
//        every arg-min, every centroid sum / mean / blend, the RMS delta partials
//   host the rand-0.8.5 stream (rng.hpp) and the decisions drawn from it, and the strictly sequential
//        f32 prefix sum of WeightedIndex (inherently serial; it consumes distances produced on the GPU)
//
// assign_points_hierarchical (k > 100) is literally a 2-level IVF search of the centroid
// table: coarse = meta-centroids (top-3, stable order), lists = centroids grouped by
// meta-centroid, k = 1 — so it runs on the same scan/select kernels as search, in LANES order.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <cstdio>
#include <memory>
#include <mutex>
#include <vector>

#include "assign_mfma.hpp"
#include "device_index.hpp"
#include "device_math.hpp"
#include "kmeans.hpp"
#include "rng.hpp"
#include "scan.hpp"

namespace vi {
namespace {

constexpr uint64_t kAssignChunk = 1u << 20;  // queries per scan launch

// ------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------
__global__ void gather_rows_kernel(const float *X, const uint32_t *idx, uint32_t nb, uint32_t d, float *out) {
  const uint32_t r = blockIdx.x;
  if (r >= nb) return;
  const float *src = X + (size_t)idx[r] * d;
  for (uint32_t j = threadIdx.x; j < d; j += blockDim.x) out[(size_t)r * d + j] = src[j];
}

// K = 1 runs: label = arg-min over the S partial winners, ties -> lower centroid index
// (find_nearest_centroid's strict '<', kmeans.rs:364-370)
__global__ void argmin_runs_kernel(const float *run_dist, const uint32_t *run_pos, uint32_t nq, uint32_t S,
                                   uint32_t *label, float *dist) {
  const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= nq) return;
  float bd = INFINITY;
  uint32_t bp = kNoPos;
  for (uint32_t s = 0; s < S; ++s) {
    const float d = run_dist[(size_t)q * S + s];
    const uint32_t p = run_pos[(size_t)q * S + s];
    if (p != kNoPos && (d < bd || (d == bd && p < bp))) { bd = d; bp = p; }
  }
  // all-infinite / NaN rows keep best_c = 0 (kmeans.rs:360)
  label[q] = bp == kNoPos ? 0u : bp;
  if (dist) dist[q] = bd;
}

// update_min_distances_parallel (kmeans.rs:422-443): rows 0..m against one centroid
__global__ void min_dist_update_kernel(const float *X, uint32_t m, uint32_t d, const float *c, float *min_d) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  const float dist = l2sq_lanes_dev(X + (size_t)i * d, c, d);
  if (dist < min_d[i]) min_d[i] = dist;
}

// sum over the members order[b..e) of X[member, j], added strictly in that order (the reference's sequential f32 sum,
// kmeans.rs:693-697).  The chain of adds is short (4 cycles each); what a big cluster costs is the latency of its row
// loads — so 32 rows are requested ahead of the 32 being added (the member ids are wave-uniform: scalar loads).
constexpr int kSegAhead = 32;
__device__ __forceinline__ float segment_sum_ordered(const float *__restrict__ X, const uint32_t *__restrict__ order,
                                                     uint32_t b, uint32_t e, uint32_t d, uint32_t j) {
  float sum = 0.0f;
  float cur[kSegAhead], nxt[kSegAhead];
#pragma unroll
  for (int u = 0; u < kSegAhead; ++u) cur[u] = b + u < e ? X[(size_t)order[b + u] * d + j] : 0.0f;
  for (uint32_t i = b; i < e; i += kSegAhead) {
    const uint32_t i2 = i + kSegAhead;
#pragma unroll
    for (int u = 0; u < kSegAhead; ++u) nxt[u] = i2 + u < e ? X[(size_t)order[i2 + u] * d + j] : 0.0f;
    if (i2 <= e) {
#pragma unroll
      for (int u = 0; u < kSegAhead; ++u) sum += cur[u];
    } else {
#pragma unroll
      for (int u = 0; u < kSegAhead; ++u) if (i + u < e) sum += cur[u];
    }
#pragma unroll
    for (int u = 0; u < kSegAhead; ++u) cur[u] = nxt[u];
  }
  return sum;
}

// What a cluster's sum becomes: kSegSums = the sum itself and the member count (per-rank partial of the data-parallel
// update), kSegMean = sum / count with zeros for an empty cluster (update_centroids_parallel, kmeans.rs:699-712),
// kSegMeanKeep = the same but an empty cluster keeps its row (build_centroid_hierarchy, kmeans.rs:634-638).
enum SegMode : int { kSegSums = 0, kSegMean = 1, kSegMeanKeep = 2 };

__device__ __forceinline__ void segment_store(int mode, float sum, uint32_t members, float *out) {
  if (mode == kSegSums) *out = sum;
  else if (members) *out = sum / (float)members;
  else if (mode == kSegMean) *out = 0.0f;
}

// one workgroup per cluster, threads over the dimensions; clusters above skip_above members are left to
// segment_big_kernel
__global__ void segment_kernel(const float *__restrict__ X, const uint32_t *__restrict__ order,
                               const uint32_t *__restrict__ seg_off, uint32_t k, uint32_t d, float *out, uint32_t *counts,
                               int mode, uint32_t skip_above) {
  const uint32_t c = blockIdx.x;
  if (c >= k) return;
  const uint32_t b = seg_off[c], e = seg_off[c + 1];
  if (e - b > skip_above) return;
  for (uint32_t j = threadIdx.x; j < d; j += blockDim.x)
    segment_store(mode, e > b ? segment_sum_ordered(X, order, b, e, d, j) : 0.0f, e - b, out + (size_t)c * d + j);
  if (counts && threadIdx.x == 0) counts[c] = e - b;
}

// A few clusters hold a large share of the points (C3: 446 of 16 384 hold 43 %, the largest 57 744) and the sum of one
// cluster is a sequential chain: with one or two waves per cluster the pass ends in a long tail of single waves waiting
// on their row loads.  Clusters above kBigCluster members are therefore summed by a whole 512-thread workgroup: all 8
// waves fetch the rows of the next chunk (kBigChunkFloats / d rows, 64 KB) while the first d threads add the staged
// chunk from LDS in member order — the order of the adds is unchanged, only who waits for memory.
constexpr uint32_t kBigCluster = 2048, kBigThreads = 512, kBigChunkFloats = 16384, kBigMaxDim = 512;
constexpr int kBigLoads = kBigChunkFloats / 4 / kBigThreads;  // 16-byte row pieces a thread requests per chunk

__global__ void segment_big_list_kernel(const uint32_t *seg_off, uint32_t k, uint32_t *nbig, uint32_t *list) {
  const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < k && seg_off[c + 1] - seg_off[c] > kBigCluster) list[atomicAdd(nbig, 1u)] = c;  // (any order: one block each)
}

typedef float vf4 __attribute__((ext_vector_type(4)));  // (arrays of HIP's float4 struct stay in scratch memory)

__global__ void __launch_bounds__(kBigThreads) __attribute__((amdgpu_waves_per_eu(2, 2))) segment_big_kernel(const float *__restrict__ X, const uint32_t *__restrict__ order,
                                                                  const uint32_t *__restrict__ seg_off,
                                                                  const uint32_t *__restrict__ nbig_ptr,
                                                                  const uint32_t *__restrict__ list, uint32_t d, float *out,
                                                                  uint32_t *counts, int mode) {
  extern __shared__ vf4 ring4[];  // two halves of kBigChunkFloats (H rows x d floats, rounded up)
  const uint32_t dq = d / 4;         // (d % 4 == 0: rows are read as 16-byte pieces — one instruction moves 1 KB; with
                                     //  4-byte pieces the CU's vector-memory issue rate, not HBM, set the pace: 3.3 us a chunk)
  const uint32_t H = kBigChunkFloats / d, t = threadIdx.x;
  const uint32_t nbig = *nbig_ptr;
  const uint32_t step_row = kBigThreads / dq, step_col = kBigThreads % dq, row0 = t / dq, col0 = t % dq;
  float *ring = reinterpret_cast<float *>(ring4);
  for (uint32_t bi = blockIdx.x; bi < nbig; bi += gridDim.x) {
    const uint32_t c = list[bi], b = seg_off[c], e = seg_off[c + 1];
    // Rows r0 .. r0 + H of the cluster into a register set, in two steps a pipeline stage apart: the member ids, then
    // the rows they name.  Piece t + p * 512 of a chunk is (row, col) = divmod(., d / 4), stepped without dividing.
    // No predication: a row past the cluster's end is clamped to its last member (staged but never added), so the
    // loads of either step are issued back to back (as conditional loads the compiler emitted dependent round trips).
    // (macros, not lambdas over array references: those left the register sets in scratch memory)
#define VI_FETCH_IDS(id, r0_)                                        \
    {                                                                \
      uint32_t row_ = row0, cl_ = col0;                              \
      _Pragma("unroll") for (int p = 0; p < kBigLoads; ++p) {        \
        id[p] = order[min((r0_) + row_, e - 1)];                     \
        row_ += step_row; cl_ += step_col;                           \
        if (cl_ >= dq) { cl_ -= dq; ++row_; }                        \
      }                                                              \
    }
#define VI_FETCH_ROWS(regs, id)                                      \
    {                                                                \
      uint32_t cl_ = col0;                                           \
      _Pragma("unroll") for (int p = 0; p < kBigLoads; ++p) {        \
        regs[p] = reinterpret_cast<const vf4 *>(X + (size_t)id[p] * d)[cl_]; \
        cl_ += step_col;                                             \
        if (cl_ >= dq) cl_ -= dq;                                    \
      }                                                              \
    }
    // (a half has room for all 4096 pieces)
#define VI_STAGE(half, regs) \
    { _Pragma("unroll") for (int p = 0; p < kBigLoads; ++p) (half)[t + p * kBigThreads] = regs[p]; }
    float sum = 0.0f;
    auto add_chunk = [&](const float *half, uint32_t r0) {  // the staged rows r0 .. r0 + H, in member order
      if (t >= d) return;
      half += t;
      const uint32_t rows = r0 < e ? min(H, e - r0) : 0u;
      uint32_t r = 0;
      for (; r + 16 <= rows; r += 16) {  // 16 LDS reads in flight, then the 16 adds
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = half[(r + u) * d];
#pragma unroll
        for (int u = 0; u < 16; ++u) sum += v[u];
      }
      for (; r < rows; ++r) sum += half[r * d];
    };
    // Three register sets rotate: while chunk n is added from LDS, chunk n+1 is copied from its set into the other half
    // and chunks n+2, n+3 are in flight.
    vf4 ra[kBigLoads], rb[kBigLoads], rc[kBigLoads];
    uint32_t idq[kBigLoads];  // ids of the chunk whose rows are requested next
    VI_FETCH_IDS(idq, b)
    VI_FETCH_ROWS(ra, idq)
    VI_FETCH_IDS(idq, b + H)
    VI_FETCH_ROWS(rb, idq)
    VI_FETCH_IDS(idq, b + 2 * H)
    VI_FETCH_ROWS(rc, idq)
    VI_FETCH_IDS(idq, b + 3 * H)
    VI_STAGE(ring4, ra)
    __syncthreads();
    uint32_t r0 = b, cur = 0;
    // one step: ring[cur] holds chunk r0, `nxt` holds chunk r0 + H, `free` is refilled with chunk r0 + 3H (its ids
    // were requested a step ago), the ids of chunk r0 + 4H are requested
    // (straight-line code, three steps per trip, no condition inside a step: with `if (r0 < e)` around each step the
    //  compiler rotated the sets by copies and waited for every load right after issuing it.  Steps past the end add
    //  nothing and fetch the clamped last row.)
#define VI_BIG_STEP(nxt, free)                         \
    {                                                  \
      add_chunk(ring + cur * kBigChunkFloats, r0);     \
      VI_FETCH_ROWS(free, idq)                         \
      VI_FETCH_IDS(idq, r0 + 4 * H)                    \
      VI_STAGE(ring4 + (cur ^ 1u) * (kBigChunkFloats / 4), nxt) \
      __syncthreads();                                 \
      cur ^= 1u; r0 += H;                              \
    }
    while (r0 < e) {
      VI_BIG_STEP(rb, ra)
      VI_BIG_STEP(rc, rb)
      VI_BIG_STEP(ra, rc)
    }
#undef VI_BIG_STEP
#undef VI_FETCH_IDS
#undef VI_FETCH_ROWS
#undef VI_STAGE
    if (t < d) segment_store(mode, sum, e - b, out + (size_t)c * d + t);
    if (counts && t == 0) counts[c] = e - b;
  }
}

struct SegWs {
  DevBuf<uint32_t> big;  // [0] = number of big clusters, then their ids
};

// Everything an update pass allocates (ids grouped by cluster, offsets, the sort's scratch, the big-cluster list): 16 n
// bytes.  A Lloyd loop keeps one across its iterations; the one-pass entry point (vi_kmeans_partial_sums_device) borrows
// one from a process-wide pool, so that a caller's training loop does not pay five allocations and releases (1.5 ms —
// as long as the pass itself at C3) per call.  Pooled workspaces are never freed (16 n bytes of 288 GB stay reserved).
struct UpdateWs {
  int device = -1;
  DevBuf<uint32_t> order, seg, scratch, bad;
  SegWs seg_ws;
};
struct UpdateWsPool {
  std::mutex m;
  std::vector<UpdateWs *> idle;
  UpdateWs *acquire(int device) {
    std::lock_guard<std::mutex> g(m);
    for (size_t i = 0; i < idle.size(); ++i)
      if (idle[i]->device == device) { UpdateWs *w = idle[i]; idle.erase(idle.begin() + i); return w; }
    UpdateWs *w = new UpdateWs;
    w->device = device;
    return w;
  }
  void release(UpdateWs *w) { std::lock_guard<std::mutex> g(m); idle.push_back(w); }
};
UpdateWsPool &update_ws_pool() {
  static UpdateWsPool *pool = new UpdateWsPool;  // (never destroyed: no hipFree after the runtime has shut down)
  return *pool;
}
struct UpdateWsLease {
  UpdateWs *ws;
  explicit UpdateWsLease(int device) : ws(update_ws_pool().acquire(device)) {}
  ~UpdateWsLease() { update_ws_pool().release(ws); }
};

// sums / means of all k clusters from the grouped ids (order, seg_off); the work is queued on st
vi_status launch_segment_sums(const float *X, const uint32_t *order, const uint32_t *seg_off, uint64_t k, uint32_t d,
                              float *out, uint32_t *counts, SegMode mode, SegWs &ws, hipStream_t st) {
  const uint32_t threads = std::min<uint32_t>(256u, (d + 63u) & ~63u);
  const bool split = d <= kBigMaxDim && d >= 16 && d % 4 == 0 && ((uintptr_t)X & 15) == 0;
  if (split) {
    VI_TRY(ws.big.reserve(k + 1));
    VI_HIP(hipMemsetAsync(ws.big.p, 0, 4, st));
    hipLaunchKernelGGL(segment_big_list_kernel, dim3((uint32_t)((k + 255) / 256)), dim3(256), 0, st, seg_off, (uint32_t)k,
                       ws.big.p, ws.big.p + 1);
  }
  hipLaunchKernelGGL(segment_kernel, dim3((uint32_t)k), dim3(threads), 0, st, X, order, seg_off, (uint32_t)k, d, out, counts,
                     (int)mode, split ? kBigCluster : 0xFFFFFFFFu);
  if (split) {
    const size_t lds = 2ull * kBigChunkFloats * sizeof(float);
    VI_HIP(hipFuncSetAttribute((const void *)segment_big_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(segment_big_kernel, dim3((uint32_t)std::min<uint64_t>(k, 1024)), dim3(kBigThreads), lds, st, X, order,
                       seg_off, ws.big.p, ws.big.p + 1, d, out, counts, (int)mode);
  }
  VI_HIP(hipGetLastError());
  return VI_OK;
}

// update_centroids_mini_batch (kmeans.rs:729-787) for the clusters touched by this batch
__global__ void minibatch_update_kernel(const float *X, const uint32_t *members, const uint32_t *t_cluster,
                                        const uint32_t *t_start, const uint32_t *t_len, const float *t_eta,
                                        uint32_t nt, uint32_t d, float *C) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (uint64_t)nt * d) return;
  const uint32_t ti = (uint32_t)(t / d), j = (uint32_t)(t % d);
  const uint32_t c = t_cluster[ti], b = t_start[ti], n = t_len[ti];
  float sum = 0.0f;
  for (uint32_t i = 0; i < n; ++i) sum += X[(size_t)members[b + i] * d + j];
  const float eta = t_eta[ti];
  const float mean = sum / (float)n;
  const float cur = C[(size_t)c * d + j];
  C[(size_t)c * d + j] = (1.0f - eta) * cur + eta * mean;
}

__global__ void copy_rows_kernel(const float *X, const uint32_t *dst_c, const uint32_t *src_row, uint32_t np,
                                 uint32_t d, float *C) {
  const uint32_t r = blockIdx.x;
  if (r >= np) return;
  const float *src = X + (size_t)src_row[r] * d;
  float *dst = C + (size_t)dst_c[r] * d;
  for (uint32_t j = threadIdx.x; j < d; j += blockDim.x) dst[j] = src[j];
}

// compute_centroid_delta (kmeans.rs:334-351): per-cluster sequential partial
__global__ void delta_kernel(const float *cur, const float *prev, uint32_t k, uint32_t d, float *local) {
  const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= k) return;
  float acc = 0.0f;
  for (uint32_t j = 0; j < d; ++j) {
    const float diff = cur[(size_t)c * d + j] - prev[(size_t)c * d + j];
    acc += diff * diff;
  }
  local[c] = acc;
}

__global__ void label_dist_kernel(const float *X, const float *C, const uint32_t *label, uint64_t n, uint32_t d,
                                  float *out) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  out[i] = l2sq_lanes_dev(X + i * d, C + (size_t)label[i] * d, d);
}

__global__ void scatter_labels_kernel(const uint32_t *rows, const uint32_t *vals, uint32_t n, uint32_t *labels) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) labels[rows[i]] = vals[i];
}

__global__ void i64_to_u32_kernel(const int64_t *in, uint64_t n, uint32_t *out) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[i] < 0 ? 0u : (uint32_t)in[i];
}

// ChaCha12 keystream in bulk (rand_chacha 0.3.1 as rng.hpp restates it): thread = one 16-word block of counter
// first + t.  sample_batch's full shuffle (kmeans.rs:722-726) reads ~1.7 n words per iteration; the host only walks them.
__global__ void chacha12_blocks_kernel(const uint32_t k0, const uint32_t k1, const uint32_t k2, const uint32_t k3,
                                       const uint32_t k4, const uint32_t k5, const uint32_t k6, const uint32_t k7,
                                       uint64_t first, uint64_t nblocks, uint32_t *out) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nblocks) return;
  const uint64_t ctr = first + t;
  const uint32_t s[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u, k0, k1, k2, k3, k4, k5, k6, k7,
                          (uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u};
  uint32_t x[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) x[i] = s[i];
#define VI_QR(a, b, c, d)                                                     \
  x[a] += x[b]; x[d] ^= x[a]; x[d] = (x[d] << 16) | (x[d] >> 16);             \
  x[c] += x[d]; x[b] ^= x[c]; x[b] = (x[b] << 12) | (x[b] >> 20);             \
  x[a] += x[b]; x[d] ^= x[a]; x[d] = (x[d] << 8) | (x[d] >> 24);              \
  x[c] += x[d]; x[b] ^= x[c]; x[b] = (x[b] << 7) | (x[b] >> 25);
#pragma unroll
  for (int r = 0; r < 6; ++r) {
    VI_QR(0, 4, 8, 12) VI_QR(1, 5, 9, 13) VI_QR(2, 6, 10, 14) VI_QR(3, 7, 11, 15)
    VI_QR(0, 5, 10, 15) VI_QR(1, 6, 11, 12) VI_QR(2, 7, 8, 13) VI_QR(3, 4, 9, 14)
  }
#undef VI_QR
  uint4 *dst = reinterpret_cast<uint4 *>(out + t * 16);
#pragma unroll
  for (int i = 0; i < 4; ++i) dst[i] = make_uint4(x[4 * i] + s[4 * i], x[4 * i + 1] + s[4 * i + 1], x[4 * i + 2] + s[4 * i + 2], x[4 * i + 3] + s[4 * i + 3]);
}

// ------------------------------------------------------------------------------------------
// device context
// ------------------------------------------------------------------------------------------
struct Ctx {
  int device = 0;
  hipStream_t st = nullptr;
  ~Ctx() { if (st) (void)hipStreamDestroy(st); }
  vi_status init(int dev) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
      return fail(VI_ERR_DEVICE, "no HIP device visible: libvi_amd never falls back to the CPU");
    if (dev < 0 || dev >= ndev) return fail(VI_ERR_DEVICE, "device %d out of range (%d visible)", dev, ndev);
    device = dev;
    VI_HIP(hipSetDevice(dev));
    VI_HIP(hipStreamCreateWithFlags(&st, hipStreamDefault));
    return VI_OK;
  }
};

// KeystreamSource (rng.hpp) on the device: blocks generated into HBM, one copy into pinned host memory
struct GpuKeystream : KeystreamSource {
  hipStream_t st;
  DevBuf<uint32_t> dev;
  uint32_t *host = nullptr;
  uint64_t host_cap = 0;
  explicit GpuKeystream(hipStream_t s) : st(s) {}
  ~GpuKeystream() override { if (host) (void)hipHostFree(host); }
  const uint32_t *blocks(const uint32_t key[8], uint64_t first_block, uint64_t nblocks) override {
    const uint64_t words = nblocks * 16;
    if (dev.reserve(words) != VI_OK) return nullptr;
    if (host_cap < words) {
      if (host) (void)hipHostFree(host);
      host = nullptr; host_cap = 0;
      if (hipHostMalloc((void **)&host, words * 4) != hipSuccess) return nullptr;
      host_cap = words;
    }
    hipLaunchKernelGGL(chacha12_blocks_kernel, dim3((uint32_t)((nblocks + 255) / 256)), dim3(256), 0, st, key[0], key[1],
                       key[2], key[3], key[4], key[5], key[6], key[7], first_block, nblocks, dev.p);
    if (hipGetLastError() != hipSuccess) return nullptr;
    if (hipMemcpyAsync(host, dev.p, words * 4, hipMemcpyDeviceToHost, st) != hipSuccess) return nullptr;
    if (hipStreamSynchronize(st) != hipSuccess) return nullptr;
    return host;
  }
};

// VI_KMEANS_TIMING=1: phase times of the training loops on stderr
bool kmeans_timing() {
  static const bool on = [] { const char *e = getenv("VI_KMEANS_TIMING"); return e && *e == '1'; }();
  return on;
}
double wall_ms() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

template <typename T>
vi_status to_device(DevBuf<T> &buf, const T *host, size_t n, hipStream_t st) {
  VI_TRY(buf.reserve(n));
  if (n) VI_HIP(hipMemcpyAsync(buf.p, host, n * sizeof(T), hipMemcpyHostToDevice, st));
  return VI_OK;
}

// stable grouping of 0..n-1 by label (ascending id inside a group) — index bookkeeping only
void group_by_label(const uint32_t *labels, uint64_t n, uint64_t k, std::vector<uint32_t> &order,
                    std::vector<uint32_t> &seg_off) {
  seg_off.assign(k + 1, 0);
  for (uint64_t i = 0; i < n; ++i) seg_off[labels[i] + 1]++;
  for (uint64_t c = 0; c < k; ++c) seg_off[c + 1] += seg_off[c];
  order.resize(n);
  std::vector<uint32_t> cur(seg_off.begin(), seg_off.end() - 1);
  for (uint64_t i = 0; i < n; ++i) order[cur[labels[i]]++] = (uint32_t)i;
}

// ------------------------------------------------------------------------------------------
// Where the training loops get data rows from.  k-means++ seeding, the mini-batches and the empty-cluster re-seeds
// touch only a few rows, named by the rand stream: 50 000 + k + iterations x (batch + empty clusters).  Reading them
// through this interface is what lets the same loop run on one GPU (rows gathered from resident X) and on data
// SHARDED over the GPUs of a node (rows exchanged by the caller's collective, vi_row_source), where only the final
// assignment of all points — the part that is O(N k D) — stays local to every rank.
// ------------------------------------------------------------------------------------------
struct RowSource {
  virtual ~RowSource() = default;
  // out_dev[i, :] = data row rows[i]; complete when the call returns or ordered on `st`
  virtual vi_status fetch(const uint32_t *rows, uint64_t n, float *out_dev, hipStream_t st) = 0;
  // rows 0 .. m-1 as one device matrix if they are resident that way (else nullptr: fetch them)
  virtual const float *head(uint64_t /*m*/) { return nullptr; }
};

struct DeviceRows : RowSource {
  const float *X;
  uint32_t d;
  DevBuf<uint32_t> idx;
  DeviceRows(const float *x, uint32_t dim) : X(x), d(dim) {}
  vi_status fetch(const uint32_t *rows, uint64_t n, float *out_dev, hipStream_t st) override {
    if (n == 0) return VI_OK;
    VI_TRY(idx.reserve(n));
    VI_HIP(hipMemcpyAsync(idx.p, rows, n * 4, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(gather_rows_kernel, dim3((uint32_t)n), dim3(64), 0, st, X, idx.p, (uint32_t)n, d, out_dev);
    VI_HIP(hipGetLastError());
    VI_HIP(hipStreamSynchronize(st));  // `rows` is the caller's host memory
    return VI_OK;
  }
  const float *head(uint64_t) override { return X; }
};

struct CallbackRows : RowSource {
  vi_row_source cb;
  std::vector<uint64_t> rows64;
  explicit CallbackRows(const vi_row_source &c) : cb(c) {}
  vi_status fetch(const uint32_t *rows, uint64_t n, float *out_dev, hipStream_t st) override {
    if (n == 0) return VI_OK;
    rows64.assign(rows, rows + n);
    VI_HIP(hipStreamSynchronize(st));  // out_dev may still be read by work queued on our stream
    const int rc = cb.fetch_rows(cb.ctx, rows64.data(), n, out_dev);
    if (rc != 0) return fail(VI_ERR_OTHER, "vi_row_source.fetch_rows failed with %d", rc);
    return VI_OK;
  }
};

// ------------------------------------------------------------------------------------------
// assign_points_brute_force on device-resident points (exact, LANES order)
// ------------------------------------------------------------------------------------------
struct BruteWs {
  DevBuf<float> cblocks, run_dist;
  DevBuf<uint32_t> ros, run_pos;
};

vi_status assign_brute_device(Ctx &cx, const float *Xq, uint64_t n, const float *Cd, uint64_t k, uint32_t d,
                              uint32_t *labels_dev, float *dist_dev, BruteWs &ws) {
  if (n == 0) return VI_OK;
  const uint32_t dq = layout_dq(d);
  const uint64_t nb = (k + 63) / 64;
  VI_TRY(ws.cblocks.reserve(std::max<uint64_t>(1, nb) * dq * 64 * 4));
  {
    std::vector<uint32_t> ros(nb * 64, kNoPos);
    for (uint64_t i = 0; i < k; ++i) ros[i] = (uint32_t)i;
    VI_TRY(to_device(ws.ros, ros.data(), ros.size(), cx.st));
    VI_TRY(launch_repack_rows(Cd, d, dq, ws.ros.p, ros.size(), nullptr, ws.cblocks.p, nullptr, cx.st));
    VI_HIP(hipStreamSynchronize(cx.st));  // ros (host vector) must outlive the copy
  }
  for (uint64_t q0 = 0; q0 < n; q0 += kAssignChunk) {
    const uint64_t nq = std::min<uint64_t>(kAssignChunk, n - q0);
    const int qg = pick_qg(dq, (double)nq, VI_ORDER_LANES);
    uint32_t bps = 0;
    const uint32_t S = coarse_splits(nq, qg, (uint32_t)nb, &bps);
    VI_TRY(ws.run_dist.reserve(nq * S));
    VI_TRY(ws.run_pos.reserve(nq * S));
    ScanArgs a{};
    a.blocks = (const float4 *)ws.cblocks.p; a.dq = dq; a.dim = d; a.Q = Xq + q0 * d; a.nq = (uint32_t)nq;
    a.K = 1; a.run_dist = ws.run_dist.p; a.run_pos = ws.run_pos.p;
    a.nvec = (uint32_t)k; a.S = S; a.bps = bps;
    const uint32_t nqg = (uint32_t)((nq + qg - 1) / qg);
    VI_TRY(launch_scan(a, qg, VI_ORDER_LANES, true, nqg * S, cx.st));
    hipLaunchKernelGGL(argmin_runs_kernel, dim3((uint32_t)((nq + 255) / 256)), dim3(256), 0, cx.st, ws.run_dist.p,
                       ws.run_pos.p, (uint32_t)nq, S, labels_dev + q0, dist_dev ? dist_dev + q0 : nullptr);
    VI_HIP(hipGetLastError());
  }
  return VI_OK;
}

// ------------------------------------------------------------------------------------------
// assign_points_hierarchical (kmeans.rs:474-581) on the device
// ------------------------------------------------------------------------------------------
vi_status assign_hier_device(Ctx &cx, const float *Xd, uint64_t n, const float *Cd, uint64_t k, uint32_t d,
                             uint64_t seed, uint32_t *labels_dev) {
  uint64_t meta_k = (uint64_t)std::sqrt((float)k);  // kmeans.rs:483
  meta_k = std::min<uint64_t>(std::max<uint64_t>(meta_k, 2), k / 2);
  const uint64_t hseed = seed * 17ULL + 42ULL;      // kmeans.rs:494
  // build_centroid_hierarchy (kmeans.rs:584-648)
  StdRng rng(hseed);
  std::vector<uint64_t> chosen = rng.choose_multiple_range(k, meta_k);
  std::vector<uint32_t> chosen32(chosen.begin(), chosen.end());
  DevBuf<uint32_t> d_idx, d_c2m, d_order, d_seg;
  DevBuf<float> meta;
  VI_TRY(meta.reserve(meta_k * d));
  VI_HIP(hipMemsetAsync(meta.p, 0, meta_k * d * sizeof(float), cx.st));
  VI_TRY(to_device(d_idx, chosen32.data(), chosen32.size(), cx.st));
  if (!chosen32.empty()) {
    hipLaunchKernelGGL(gather_rows_kernel, dim3((uint32_t)chosen32.size()), dim3(64), 0, cx.st, Cd, d_idx.p,
                       (uint32_t)chosen32.size(), d, meta.p);
    VI_HIP(hipGetLastError());
  }
  VI_TRY(d_c2m.reserve(k));
  std::vector<uint32_t> c2m(k), order, seg;
  BruteWs bws;
  SegWs sws;
  for (int iter = 0; iter < 5; ++iter) {
    VI_TRY(assign_brute_device(cx, Cd, k, meta.p, meta_k, d, d_c2m.p, nullptr, bws));
    VI_HIP(hipMemcpyAsync(c2m.data(), d_c2m.p, k * 4, hipMemcpyDeviceToHost, cx.st));
    VI_HIP(hipStreamSynchronize(cx.st));
    group_by_label(c2m.data(), k, meta_k, order, seg);
    VI_TRY(to_device(d_order, order.data(), order.size(), cx.st));
    VI_TRY(to_device(d_seg, seg.data(), seg.size(), cx.st));
    VI_TRY(launch_segment_sums(Cd, d_order.p, d_seg.p, meta_k, d, meta.p, nullptr, kSegMeanKeep, sws, cx.st));
    VI_HIP(hipStreamSynchronize(cx.st));
  }
  // two-level index: coarse = meta-centroids, list m = centroids of meta cluster m, ascending c
  // (meta_to_centroids, kmeans.rs:518-521); order/seg hold exactly that grouping.
  std::vector<uint64_t> list_off(seg.begin(), seg.end());
  DeviceIndex hix;
  VI_TRY(device_index_from_rows(cx.device, VI_ORDER_LANES, d, meta.p, meta_k, Cd, list_off, order, nullptr, nullptr,
                                &hix));
  const uint64_t top = std::min<uint64_t>(3, meta_k);  // kmeans.rs:536
  DevBuf<float> Dd;
  DevBuf<int64_t> Id;
  for (uint64_t q0 = 0; q0 < n; q0 += kAssignChunk) {
    const uint64_t nq = std::min<uint64_t>(kAssignChunk, n - q0);
    VI_TRY(Dd.reserve(nq));
    VI_TRY(Id.reserve(nq));
    SearchIO io;
    io.queries = Xd + q0 * d; io.on_device = true; io.nq = nq; io.k = 1; io.n_probe = top;
    io.D = Dd.p; io.I = Id.p;
    VI_TRY(device_index_search(hix, io));
    VI_HIP(hipSetDevice(cx.device));
    hipLaunchKernelGGL(i64_to_u32_kernel, dim3((uint32_t)((nq + 255) / 256)), dim3(256), 0, cx.st, Id.p, nq,
                       labels_dev + q0);
    VI_HIP(hipGetLastError());
    VI_HIP(hipStreamSynchronize(cx.st));
  }
  return VI_OK;
}

// exact re-evaluation of the rows the MFMA filter could not decide (assign_mfma.hpp: ExactRowsFn)
struct ExactCtx {
  Ctx *cx;
  const float *Cd;
  uint64_t k;
  uint32_t d;
  BruteWs *bws;
  DevBuf<float> rows;
  DevBuf<uint32_t> labels;
};

vi_status exact_rows_cb(void *vctx, const float *X, const uint32_t *rows_dev, uint32_t nrows, uint32_t *labels) {
  ExactCtx &e = *static_cast<ExactCtx *>(vctx);
  VI_TRY(e.rows.reserve((uint64_t)nrows * e.d));
  VI_TRY(e.labels.reserve(nrows));
  hipLaunchKernelGGL(gather_rows_kernel, dim3(nrows), dim3(64), 0, e.cx->st, X, rows_dev, nrows, e.d, e.rows.p);
  VI_HIP(hipGetLastError());
  VI_TRY(assign_brute_device(*e.cx, e.rows.p, nrows, e.Cd, e.k, e.d, e.labels.p, nullptr, *e.bws));
  hipLaunchKernelGGL(scatter_labels_kernel, dim3((nrows + 255) / 256), dim3(256), 0, e.cx->st, rows_dev, e.labels.p,
                     nrows, labels);
  VI_HIP(hipGetLastError());
  VI_HIP(hipStreamSynchronize(e.cx->st));
  return VI_OK;
}

// exact brute-force assignment: MFMA filter + exact re-check where the shape allows, else the
// exact-order scan kernel alone.  Both give assign_points_brute_force's labels bit for bit.
vi_status assign_exact_device(Ctx &cx, const float *Xd, uint64_t n, const float *Cd, uint64_t k, uint32_t d,
                              uint32_t *labels_dev, BruteWs &bws, MfmaAssignStats *stats = nullptr) {
  const char *off = getenv("VI_NO_MFMA");
  if (mfma_assign_supported(n, k, d) && !(off && *off == '1')) {
    MfmaAssignWs mws;
    ExactCtx ectx{&cx, Cd, k, d, &bws, {}, {}};
    return mfma_assign_device(Xd, n, Cd, k, d, labels_dev, mws, cx.st, exact_rows_cb, &ectx, stats);
  }
  return assign_brute_device(cx, Xd, n, Cd, k, d, labels_dev, nullptr, bws);
}

// assign_points_simd_parallel (kmeans.rs:445-459)
vi_status assign_device(Ctx &cx, const float *Xd, uint64_t n, const float *Cd, uint64_t k, uint32_t d, uint64_t seed,
                        vi_assign_mode mode, uint32_t *labels_dev, BruteWs &bws) {
  if (mode == VI_ASSIGN_REFERENCE && k > 100) return assign_hier_device(cx, Xd, n, Cd, k, d, seed, labels_dev);
  return assign_exact_device(cx, Xd, n, Cd, k, d, labels_dev, bws);
}

// ------------------------------------------------------------------------------------------
// kmeans_plus_plus_init (kmeans.rs:154-310): distances on the GPU, sampling decisions on host
// ------------------------------------------------------------------------------------------
// The rows this touches are known from (n, seed) alone: the first centroid's row, the rows 0..m-1 that are measured
// (m = min(n, 50 000): kmeans.rs:268,435 measures data rows 0..m, not the sampled rows) and the m candidate rows the
// draws map to (sample_indices, :287-288; the identity when n <= 50 000).
vi_status kmeans_pp_init_rows(Ctx &cx, RowSource &src, uint64_t n, uint32_t d, uint64_t k, uint64_t seed, float *Cd) {
  const uint64_t sample_threshold = 50000;
  StdRng rng(seed);
  const uint64_t actual_k = std::min(k, n);
  const uint32_t row0 = (uint32_t)rng.gen_range(0, n);
  const bool sampled = n > sample_threshold;
  std::vector<uint32_t> sample_idx;
  uint64_t m = n;
  if (sampled) {
    sample_idx.resize(n);
    for (uint64_t i = 0; i < n; ++i) sample_idx[i] = (uint32_t)i;
    rng.shuffle(sample_idx.data(), n);
    m = std::min(sample_threshold, n);
  }
  // cand: row 0 = the first centroid's data row, rows 1..m = the candidates of the draws
  DevBuf<float> cand, headbuf;
  VI_TRY(cand.reserve((m + 1) * d));
  const float *head = src.head(m);
  {
    std::vector<uint32_t> rows(m + 1);
    rows[0] = row0;
    for (uint64_t s = 0; s < m; ++s) rows[s + 1] = sampled ? sample_idx[s] : (uint32_t)s;
    VI_TRY(src.fetch(rows.data(), m + 1, cand.p, cx.st));
    if (!head) {
      if (!sampled) head = cand.p + d;  // the candidates ARE rows 0..m-1
      else {
        for (uint64_t s = 0; s < m; ++s) rows[s] = (uint32_t)s;
        VI_TRY(headbuf.reserve(m * d));
        VI_TRY(src.fetch(rows.data(), m, headbuf.p, cx.st));
        head = headbuf.p;
      }
    }
  }
  std::vector<uint32_t> csel(k, 0);  // every initial centroid is a copy of a row of `cand`
  DevBuf<float> min_d;
  VI_TRY(min_d.reserve(m));
  std::vector<float> h_min(m, INFINITY), cum(m);
  const double t_pp0 = wall_ms();
  VI_HIP(hipMemcpyAsync(min_d.p, h_min.data(), m * 4, hipMemcpyHostToDevice, cx.st));
  float *pinned = nullptr;
  VI_HIP(hipHostMalloc((void **)&pinned, std::max<uint64_t>(m, 1) * sizeof(float)));
  vi_status rc = VI_OK;
  for (uint64_t i = 1; i < actual_k && rc == VI_OK; ++i) {
    hipLaunchKernelGGL(min_dist_update_kernel, dim3((uint32_t)((m + 255) / 256)), dim3(256), 0, cx.st, head,
                       (uint32_t)m, d, cand.p + (size_t)csel[i - 1] * d, min_d.p);
    if (hipMemcpyAsync(pinned, min_d.p, m * 4, hipMemcpyDeviceToHost, cx.st) != hipSuccess ||
        hipStreamSynchronize(cx.st) != hipSuccess) {
      rc = fail(VI_ERR_DEVICE, "k-means++ distance pass failed: %s", hipGetErrorString(hipGetLastError()));
      break;
    }
    // weights = dist^4 (:190), their sum (:193) and WeightedIndex's cumulative weights are ONE left-to-right chain
    // (0 + w0 = w0 exactly): cum[j] = w0 + .. + wj
    float total = pinned[0] * pinned[0];
    for (uint64_t j = 1; j < m; ++j) { cum[j - 1] = total; total += pinned[j] * pinned[j]; }
    if (total == 0.0f) csel[i] = csel[rng.gen_range(0, i)];
    else csel[i] = (uint32_t)rng.weighted_index_cum(cum.data(), total, m) + 1u;
  }
  if (kmeans_timing()) fprintf(stderr, "[vi kmeans] k-means++ %llu draws over %llu rows: %.1f ms\n", (unsigned long long)actual_k, (unsigned long long)m, wall_ms() - t_pp0);
  (void)hipHostFree(pinned);
  VI_TRY(rc);
  for (uint64_t i = actual_k; i < k; ++i) csel[i] = csel[rng.gen_range(0, actual_k)];
  DevBuf<uint32_t> d_rows;
  VI_TRY(to_device(d_rows, csel.data(), csel.size(), cx.st));
  hipLaunchKernelGGL(gather_rows_kernel, dim3((uint32_t)k), dim3(64), 0, cx.st, cand.p, d_rows.p, (uint32_t)k, d, Cd);
  VI_HIP(hipGetLastError());
  VI_HIP(hipStreamSynchronize(cx.st));
  return VI_OK;
}

// handle_empty_clusters (kmeans.rs:313-331)
vi_status handle_empty_rows(Ctx &cx, RowSource &src, uint64_t n, uint32_t d, const std::vector<uint64_t> &counts,
                            StdRng &rng, float *Cd, DevBuf<float> &rowbuf) {
  std::vector<uint32_t> dst, rows, pos;
  for (uint64_t c = 0; c < counts.size(); ++c)
    if (counts[c] == 0) { dst.push_back((uint32_t)c); rows.push_back((uint32_t)rng.gen_range(0, n)); }
  if (dst.empty()) return VI_OK;
  VI_TRY(rowbuf.reserve(rows.size() * (size_t)d));
  VI_TRY(src.fetch(rows.data(), rows.size(), rowbuf.p, cx.st));
  pos.resize(rows.size());
  for (size_t i = 0; i < pos.size(); ++i) pos[i] = (uint32_t)i;
  DevBuf<uint32_t> d_dst, d_src;
  VI_TRY(to_device(d_dst, dst.data(), dst.size(), cx.st));
  VI_TRY(to_device(d_src, pos.data(), pos.size(), cx.st));
  hipLaunchKernelGGL(copy_rows_kernel, dim3((uint32_t)dst.size()), dim3(64), 0, cx.st, rowbuf.p, d_dst.p, d_src.p,
                     (uint32_t)dst.size(), d, Cd);
  VI_HIP(hipGetLastError());
  VI_HIP(hipStreamSynchronize(cx.st));
  return VI_OK;
}

// compute_centroid_delta (kmeans.rs:334-351).  Rayon's reduction order is unspecified in the
// reference; the per-cluster partials are summed in cluster order here (as the oracle does).
vi_status centroid_delta_device(Ctx &cx, const float *cur, const float *prev, uint64_t k, uint32_t d,
                                DevBuf<float> &local, std::vector<float> &h_local, float *delta) {
  VI_TRY(local.reserve(k));
  hipLaunchKernelGGL(delta_kernel, dim3((uint32_t)((k + 255) / 256)), dim3(256), 0, cx.st, cur, prev, (uint32_t)k, d,
                     local.p);
  VI_HIP(hipGetLastError());
  h_local.resize(k);
  VI_HIP(hipMemcpyAsync(h_local.data(), local.p, k * 4, hipMemcpyDeviceToHost, cx.st));
  VI_HIP(hipStreamSynchronize(cx.st));
  float dsq = 0.0f;
  for (uint64_t c = 0; c < k; ++c) dsq += h_local[c];
  *delta = std::sqrt(dsq / (float)(k * d));
  return VI_OK;
}

vi_status labels_to_host(Ctx &cx, const uint32_t *labels_dev, uint64_t n, uint64_t *out) {
  std::vector<uint32_t> l32(n);
  VI_HIP(hipMemcpyAsync(l32.data(), labels_dev, n * 4, hipMemcpyDeviceToHost, cx.st));
  VI_HIP(hipStreamSynchronize(cx.st));
  for (uint64_t i = 0; i < n; ++i) out[i] = l32[i];
  return VI_OK;
}

}  // namespace

// ------------------------------------------------------------------------------------------
// public entry points
// ------------------------------------------------------------------------------------------
vi_status assign_points(const float *X, uint64_t n, uint32_t d, const float *C, uint64_t k, uint64_t seed,
                        const KMeansOptions &opt, uint64_t *labels, float *dist_out) {
  Ctx cx;
  VI_TRY(cx.init(opt.device));
  DevBuf<float> Xd, Cd, dist;
  DevBuf<uint32_t> lab;
  VI_TRY(to_device(Xd, X, n * d, cx.st));
  VI_TRY(to_device(Cd, C, k * d, cx.st));
  VI_TRY(lab.reserve(n));
  BruteWs bws;
  VI_TRY(assign_device(cx, Xd.p, n, Cd.p, k, d, seed, opt.mode, lab.p, bws));
  if (dist_out) {
    VI_TRY(dist.reserve(n));
    hipLaunchKernelGGL(label_dist_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, cx.st, Xd.p, Cd.p, lab.p, n,
                       d, dist.p);
    VI_HIP(hipGetLastError());
    VI_HIP(hipMemcpyAsync(dist_out, dist.p, n * 4, hipMemcpyDeviceToHost, cx.st));
  }
  return labels_to_host(cx, lab.p, n, labels);
}

vi_status assign_points_device(int device, const float *Xd, uint64_t n, uint32_t d, const float *Cd, uint64_t k,
                               uint64_t seed, vi_assign_mode mode, uint32_t *labels_dev, vi_assign_stats *stats) {
  Ctx cx;
  VI_TRY(cx.init(device));
  BruteWs bws;
  hipEvent_t e0, e1;
  VI_HIP(hipEventCreate(&e0));
  VI_HIP(hipEventCreate(&e1));
  VI_HIP(hipEventRecord(e0, cx.st));
  MfmaAssignStats ms;
  if (stats) { ms.export_rows = stats->ambiguous_rows_dev; ms.export_cap = stats->ambiguous_cap; }
  vi_status rc;
  const bool hier = mode == VI_ASSIGN_REFERENCE && k > 100;
  if (hier) rc = assign_hier_device(cx, Xd, n, Cd, k, d, seed, labels_dev);
  else rc = assign_exact_device(cx, Xd, n, Cd, k, d, labels_dev, bws, &ms);
  if (rc == VI_OK) {
    VI_HIP(hipEventRecord(e1, cx.st));
    VI_HIP(hipStreamSynchronize(cx.st));
    if (stats) {
      stats->n = n; stats->k = k;
      stats->ambiguous_rows = ms.ambiguous_rows;
      stats->tier1_rows = ms.tier1_rows;
      stats->ms_filter = ms.ms_filter;
      stats->used_mfma = (!hier && ms.ms_filter > 0.0f) ? 1u : 0u;
      (void)hipEventElapsedTime(&stats->ms_total, e0, e1);
    }
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return rc;
}

// ------------------------------------------------------------------------------------------
// per-rank pieces of the data-parallel Lloyd update (update_centroids_parallel, kmeans.rs:674-719)
// ------------------------------------------------------------------------------------------
__global__ void label_range_kernel(const uint32_t *labels, uint64_t n, uint32_t k, uint32_t *bad) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && labels[i] >= k) atomicOr(bad, 1u);
}

// C_new[c] = sums[c] / counts[c] (zeros for an empty cluster, kmeans.rs:705-712), and the per-cluster partial of
// compute_centroid_delta against C_prev
__global__ void finish_update_kernel(const float *sums, const uint32_t *counts, const float *prev, uint32_t k, uint32_t d,
                                     float *Cn, float *local) {
  const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= k) return;
  const uint32_t cnt = counts[c];
  float acc = 0.0f;
  for (uint32_t j = 0; j < d; ++j) {
    const float v = cnt ? sums[(size_t)c * d + j] / (float)cnt : 0.0f;
    Cn[(size_t)c * d + j] = v;
    const float diff = v - prev[(size_t)c * d + j];
    acc += diff * diff;
  }
  local[c] = acc;
}

namespace {

// run_kmeans_parallel (kmeans.rs:15-60) on device-resident points
vi_status lloyd_core(Ctx &cx, const float *Xd, uint64_t n, uint32_t d, uint64_t k, uint64_t max_iters, float thr,
                     uint64_t seed, vi_assign_mode mode, float *Cd, uint32_t *lab, uint64_t *iters_run) {
  StdRng rng(seed);
  DeviceRows src(Xd, d);
  DevBuf<float> Cn, local, rowbuf;
  UpdateWs uws;
  VI_TRY(Cn.reserve(k * d));
  VI_HIP(hipMemsetAsync(lab, 0, n * 4, cx.st));
  VI_TRY(kmeans_pp_init_rows(cx, src, n, d, k, seed, Cd));
  BruteWs bws;
  std::vector<uint64_t> counts(k), off;
  std::vector<float> h_local;
  uint64_t it = 0;
  for (; it < max_iters; ++it) {
    VI_TRY(assign_device(cx, Xd, n, Cd, k, d, seed, mode, lab, bws));
    // members of every cluster in ascending id (kmeans.rs:688-697), grouped on the device (list_build.hip)
    VI_TRY(group_ids_by_label_device(lab, n, k, uws.order, uws.seg, &off, cx.st, &uws.scratch));
    VI_TRY(launch_segment_sums(Xd, uws.order.p, uws.seg.p, k, d, Cn.p, nullptr, kSegMean, uws.seg_ws, cx.st));
    for (uint64_t c = 0; c < k; ++c) counts[c] = off[c + 1] - off[c];
    VI_TRY(handle_empty_rows(cx, src, n, d, counts, rng, Cn.p, rowbuf));
    float delta = 0.0f;
    VI_TRY(centroid_delta_device(cx, Cn.p, Cd, k, d, local, h_local, &delta));
    VI_HIP(hipMemcpyAsync(Cd, Cn.p, k * d * 4, hipMemcpyDeviceToDevice, cx.st));
    if (delta < thr) { ++it; break; }
  }
  if (iters_run) *iters_run = it;
  VI_HIP(hipStreamSynchronize(cx.st));
  return VI_OK;
}

// the training loop of run_kmeans_mini_batch (kmeans.rs:64-142) without its final assignment: everything it reads
// of the data comes through `src`
vi_status mini_batch_train_core(Ctx &cx, RowSource &src, uint64_t n, uint32_t d, uint64_t k, uint64_t max_iters,
                                float thr, uint64_t seed, float *Cd, uint64_t *iters_run) {
  StdRng rng(seed);
  const uint64_t B = std::min<uint64_t>(vi_minibatch_size(n), n);  // kmeans.rs:83; take(batch) of n indices
  DevBuf<float> prev, Qb, local, d_eta, rowbuf;
  DevBuf<uint32_t> d_blab, d_members, d_tc, d_ts, d_tl;
  VI_TRY(prev.reserve(k * d));
  VI_TRY(Qb.reserve(B * d));
  VI_TRY(d_blab.reserve(B));
  VI_TRY(kmeans_pp_init_rows(cx, src, n, d, k, seed, Cd));
  VI_HIP(hipMemcpyAsync(prev.p, Cd, k * d * 4, hipMemcpyDeviceToDevice, cx.st));
  std::vector<uint64_t> counts(k, 0);
  std::vector<uint32_t> bidx(B), blab(B), members, tc, ts, tl, draws;
  std::vector<uint64_t> follow_bits;
  GpuKeystream gks(cx.st);
  double t_shuffle = 0.0;
  const double t_loop0 = wall_ms();
  std::vector<float> teta, h_local;
  std::vector<uint32_t> head(k), nxt(B);
  BruteWs bws;
  uint64_t it = 0;
  for (; it < max_iters; ++it) {
    // sample_batch (kmeans.rs:722-726): full shuffle of 0..n, first B — only those B entries are formed (rng.hpp)
    const double t_s0 = wall_ms();
    if (!rng.shuffle_head(n, B, bidx.data(), gks, draws, follow_bits)) {
      HostKeystream hks;
      if (!rng.shuffle_head(n, B, bidx.data(), hks, draws, follow_bits)) return fail(VI_ERR_OTHER, "sample_batch failed");
    }
    t_shuffle += wall_ms() - t_s0;
    VI_TRY(src.fetch(bidx.data(), B, Qb.p, cx.st));
    // batch assignment is always brute force over all k (kmeans.rs:103-110)
    VI_TRY(assign_exact_device(cx, Qb.p, B, Cd, k, d, d_blab.p, bws));
    VI_HIP(hipMemcpyAsync(blab.data(), d_blab.p, B * 4, hipMemcpyDeviceToHost, cx.st));
    VI_HIP(hipStreamSynchronize(cx.st));
    // group the batch by cluster, batch order inside a cluster (kmeans.rs:739-742); members = positions in the batch
    members.clear(); tc.clear(); ts.clear(); tl.clear(); teta.clear();
    std::fill(head.begin(), head.end(), kNoPos);
    std::vector<uint32_t> tail_of(k, kNoPos), touched;
    for (uint32_t b = 0; b < B; ++b) {
      const uint32_t c = blab[b];
      nxt[b] = kNoPos;
      if (head[c] == kNoPos) { head[c] = b; touched.push_back(c); }
      else nxt[tail_of[c]] = b;
      tail_of[c] = b;
    }
    std::sort(touched.begin(), touched.end());
    for (uint32_t c : touched) {
      tc.push_back(c);
      ts.push_back((uint32_t)members.size());
      uint32_t len = 0;
      for (uint32_t b = head[c]; b != kNoPos; b = nxt[b]) { members.push_back(b); ++len; }
      tl.push_back(len);
      counts[c] += 1;                               // per ITERATION, not per point (:757)
      teta.push_back(1.0f / (float)counts[c]);      // eta = 1/new_count (:758)
    }
    VI_TRY(to_device(d_members, members.data(), members.size(), cx.st));
    VI_TRY(to_device(d_tc, tc.data(), tc.size(), cx.st));
    VI_TRY(to_device(d_ts, ts.data(), ts.size(), cx.st));
    VI_TRY(to_device(d_tl, tl.data(), tl.size(), cx.st));
    VI_TRY(to_device(d_eta, teta.data(), teta.size(), cx.st));
    const uint64_t nt = (uint64_t)tc.size() * d;
    hipLaunchKernelGGL(minibatch_update_kernel, dim3((uint32_t)((nt + 255) / 256)), dim3(256), 0, cx.st, Qb.p,
                       d_members.p, d_tc.p, d_ts.p, d_tl.p, d_eta.p, (uint32_t)tc.size(), d, Cd);
    VI_HIP(hipGetLastError());
    VI_HIP(hipStreamSynchronize(cx.st));
    VI_TRY(handle_empty_rows(cx, src, n, d, counts, rng, Cd, rowbuf));
    float delta = 0.0f;
    VI_TRY(centroid_delta_device(cx, Cd, prev.p, k, d, local, h_local, &delta));
    VI_HIP(hipMemcpyAsync(prev.p, Cd, k * d * 4, hipMemcpyDeviceToDevice, cx.st));
    if (delta < thr) { ++it; break; }
  }
  if (iters_run) *iters_run = it;
  VI_HIP(hipStreamSynchronize(cx.st));
  if (kmeans_timing()) fprintf(stderr, "[vi kmeans] mini-batch loop %llu iterations: %.1f ms, of which sample_batch %.1f ms\n", (unsigned long long)it, wall_ms() - t_loop0, t_shuffle);
  return VI_OK;
}

vi_status check_kmeans_args(uint64_t n, uint32_t d, uint64_t k, const void *X) {
  if (n == 0 || d == 0 || !X) return fail(VI_ERR_INVALID_INPUT, "Input vectors cannot be empty");  // kmeans.rs:23-28,72-77
  if (k == 0) return fail(VI_ERR_INVALID_INPUT, "k must be greater than 0");
  if (n > 0xFFFFFFFEull) return fail(VI_ERR_INVALID_INPUT, "more than 2^32 - 2 points");
  return VI_OK;
}

}  // namespace

// run_kmeans_parallel — host-pointer form
vi_status kmeans_parallel(const float *X, uint64_t n, uint32_t d, uint64_t k, uint64_t max_iters, float thr,
                          uint64_t seed, const KMeansOptions &opt, float *C, uint64_t *labels, uint64_t *iters_run) {
  if (thr < 0) thr = 1e-4f;  // unwrap_or(1e-4), kmeans.rs:22
  VI_TRY(check_kmeans_args(n, d, k, X));
  Ctx cx;
  VI_TRY(cx.init(opt.device));
  DevBuf<float> Xd, Cd;
  DevBuf<uint32_t> lab;
  VI_TRY(to_device(Xd, X, n * d, cx.st));
  VI_TRY(Cd.reserve(k * d));
  VI_TRY(lab.reserve(n));
  VI_TRY(lloyd_core(cx, Xd.p, n, d, k, max_iters, thr, seed, opt.mode, Cd.p, lab.p, iters_run));
  VI_HIP(hipMemcpyAsync(C, Cd.p, k * d * 4, hipMemcpyDeviceToHost, cx.st));
  return labels_to_host(cx, lab.p, n, labels);
}

vi_status kmeans_parallel_device(int device, const float *Xd, uint64_t n, uint32_t d, uint64_t k, uint64_t max_iters,
                                 float thr, uint64_t seed, vi_assign_mode mode, float *Cd, uint32_t *labels_dev,
                                 uint64_t *iters_run) {
  if (thr < 0) thr = 1e-4f;
  VI_TRY(check_kmeans_args(n, d, k, Xd));
  if (!Cd || !labels_dev) return fail(VI_ERR_INVALID_INPUT, "null output pointer");
  Ctx cx;
  VI_TRY(cx.init(device));
  return lloyd_core(cx, Xd, n, d, k, max_iters, thr, seed, mode, Cd, labels_dev, iters_run);
}

// run_kmeans_mini_batch — host-pointer form
vi_status kmeans_mini_batch(const float *X, uint64_t n, uint32_t d, uint64_t k, uint64_t max_iters, float thr,
                            uint64_t seed, const KMeansOptions &opt, float *C, uint64_t *labels, uint64_t *iters_run) {
  if (thr < 0) thr = 1e-4f;  // kmeans.rs:71
  VI_TRY(check_kmeans_args(n, d, k, X));
  Ctx cx;
  VI_TRY(cx.init(opt.device));
  DevBuf<float> Xd, Cd;
  DevBuf<uint32_t> lab;
  VI_TRY(to_device(Xd, X, n * d, cx.st));
  VI_TRY(Cd.reserve(k * d));
  DeviceRows src(Xd.p, d);
  VI_TRY(mini_batch_train_core(cx, src, n, d, k, max_iters, thr, seed, Cd.p, iters_run));
  // final assignment of all points (kmeans.rs:144-147)
  VI_TRY(lab.reserve(n));
  BruteWs bws;
  VI_TRY(assign_device(cx, Xd.p, n, Cd.p, k, d, seed, opt.mode, lab.p, bws));
  VI_HIP(hipMemcpyAsync(C, Cd.p, k * d * 4, hipMemcpyDeviceToHost, cx.st));
  return labels_to_host(cx, lab.p, n, labels);
}

vi_status kmeans_mini_batch_device(int device, const float *Xd, uint64_t n, uint32_t d, uint64_t k, uint64_t max_iters,
                                   float thr, uint64_t seed, vi_assign_mode mode, float *Cd, uint32_t *labels_dev,
                                   uint64_t *iters_run) {
  if (thr < 0) thr = 1e-4f;
  VI_TRY(check_kmeans_args(n, d, k, Xd));
  if (!Cd) return fail(VI_ERR_INVALID_INPUT, "null output pointer");
  Ctx cx;
  VI_TRY(cx.init(device));
  DeviceRows src(Xd, d);
  VI_TRY(mini_batch_train_core(cx, src, n, d, k, max_iters, thr, seed, Cd, iters_run));
  if (labels_dev) {
    BruteWs bws;
    VI_TRY(assign_device(cx, Xd, n, Cd, k, d, seed, mode, labels_dev, bws));
    VI_HIP(hipStreamSynchronize(cx.st));
  }
  return VI_OK;
}

// the training loop alone over a caller-provided row source (data sharded over the GPUs of a node)
vi_status kmeans_mini_batch_train(int device, const vi_row_source &rows, uint64_t n, uint32_t d, uint64_t k,
                                  uint64_t max_iters, float thr, uint64_t seed, float *Cd, uint64_t *iters_run) {
  if (thr < 0) thr = 1e-4f;
  if (!rows.fetch_rows) return fail(VI_ERR_INVALID_INPUT, "vi_row_source.fetch_rows is null");
  VI_TRY(check_kmeans_args(n, d, k, &rows));
  if (!Cd) return fail(VI_ERR_INVALID_INPUT, "null output pointer");
  Ctx cx;
  VI_TRY(cx.init(device));
  CallbackRows src(rows);
  return mini_batch_train_core(cx, src, n, d, k, max_iters, thr, seed, Cd, iters_run);
}

vi_status kmeans_pp_init_rows_entry(int device, const vi_row_source &rows, uint64_t n, uint32_t d, uint64_t k,
                                    uint64_t seed, float *Cd) {
  if (!rows.fetch_rows) return fail(VI_ERR_INVALID_INPUT, "vi_row_source.fetch_rows is null");
  VI_TRY(check_kmeans_args(n, d, k, &rows));
  if (!Cd) return fail(VI_ERR_INVALID_INPUT, "null output pointer");
  Ctx cx;
  VI_TRY(cx.init(device));
  CallbackRows src(rows);
  return kmeans_pp_init_rows(cx, src, n, d, k, seed, Cd);
}

// per-rank partial sums / counts of the Lloyd update over this rank's points (labels from vi_assign_device)
vi_status kmeans_partial_sums_device(int device, const float *Xd, uint64_t n, uint32_t d, const uint32_t *labels_dev,
                                     uint64_t k, float *sums_dev, uint32_t *counts_dev) {
  if (d == 0 || k == 0 || !sums_dev || !counts_dev || (n && (!Xd || !labels_dev)))
    return fail(VI_ERR_INVALID_INPUT, "bad arguments to vi_kmeans_partial_sums_device");
  if (n > 0xFFFFFFFEull) return fail(VI_ERR_INVALID_INPUT, "more than 2^32 - 2 points");
  Ctx cx;
  VI_TRY(cx.init(device));
  // ids grouped by cluster, ascending id inside a cluster, without leaving the device (list_build.hip)
  UpdateWsLease lease(device);
  UpdateWs &w = *lease.ws;
  VI_TRY(w.bad.reserve(1));
  VI_HIP(hipMemsetAsync(w.bad.p, 0, 4, cx.st));
  if (n) {
    hipLaunchKernelGGL(label_range_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, cx.st, labels_dev, n, (uint32_t)k, w.bad.p);
    VI_HIP(hipGetLastError());
  }
  uint32_t bad = 0;
  VI_HIP(hipMemcpyAsync(&bad, w.bad.p, 4, hipMemcpyDeviceToHost, cx.st));
  VI_HIP(hipStreamSynchronize(cx.st));
  if (bad) return fail(VI_ERR_INVALID_INPUT, "a label is not below k");
  VI_TRY(group_ids_by_label_device(labels_dev, n, k, w.order, w.seg, nullptr, cx.st, &w.scratch));
  VI_TRY(launch_segment_sums(Xd, w.order.p, w.seg.p, k, d, sums_dev, counts_dev, kSegSums, w.seg_ws, cx.st));
  VI_HIP(hipStreamSynchronize(cx.st));
  return VI_OK;
}

// after the all-reduce of sums / counts: the new centroids, the RMS movement against the previous ones
// (compute_centroid_delta, kmeans.rs:334-351) and the clusters that received no point (kmeans.rs:313-331 re-seeds them)
vi_status kmeans_finish_update_device(int device, const float *sums_dev, const uint32_t *counts_dev, uint64_t k, uint32_t d,
                                      const float *C_prev_dev, float *C_new_dev, float *delta_out, uint32_t *empty_out,
                                      uint64_t *n_empty) {
  if (d == 0 || k == 0 || !sums_dev || !counts_dev || !C_prev_dev || !C_new_dev)
    return fail(VI_ERR_INVALID_INPUT, "bad arguments to vi_kmeans_finish_update_device");
  Ctx cx;
  VI_TRY(cx.init(device));
  DevBuf<float> local;
  VI_TRY(local.reserve(k));
  hipLaunchKernelGGL(finish_update_kernel, dim3((uint32_t)((k + 255) / 256)), dim3(256), 0, cx.st, sums_dev, counts_dev,
                     C_prev_dev, (uint32_t)k, d, C_new_dev, local.p);
  VI_HIP(hipGetLastError());
  std::vector<float> h_local(k);
  std::vector<uint32_t> h_cnt(k);
  VI_HIP(hipMemcpyAsync(h_local.data(), local.p, k * 4, hipMemcpyDeviceToHost, cx.st));
  VI_HIP(hipMemcpyAsync(h_cnt.data(), counts_dev, k * 4, hipMemcpyDeviceToHost, cx.st));
  VI_HIP(hipStreamSynchronize(cx.st));
  float dsq = 0.0f;
  for (uint64_t c = 0; c < k; ++c) dsq += h_local[c];
  if (delta_out) *delta_out = std::sqrt(dsq / (float)(k * d));
  uint64_t ne = 0;
  for (uint64_t c = 0; c < k; ++c)
    if (h_cnt[c] == 0) { if (empty_out) empty_out[ne] = (uint32_t)c; ++ne; }
  if (n_empty) *n_empty = ne;
  return VI_OK;
}

// compute_centroid_delta (kmeans.rs:334-351) of two device centroid tables
vi_status kmeans_centroid_delta(int device, const float *cur_dev, const float *prev_dev, uint64_t k, uint32_t d, float *delta) {
  if (!cur_dev || !prev_dev || !delta || k == 0 || d == 0) return fail(VI_ERR_INVALID_INPUT, "bad arguments to vi_kmeans_centroid_delta_device");
  Ctx cx;
  VI_TRY(cx.init(device));
  DevBuf<float> local;
  std::vector<float> h_local;
  return centroid_delta_device(cx, cur_dev, prev_dev, k, d, local, h_local, delta);
}

}  // namespace vi
