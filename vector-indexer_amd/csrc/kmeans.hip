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
//   host the rand-0.8.5 stream (rng.hpp) and the decisions drawn from it, stable grouping of
//        point ids by label (index bookkeeping), and the strictly sequential f32 prefix sum of
//        WeightedIndex (inherently serial; it consumes distances produced on the GPU)
//
// assign_points_hierarchical (k > 100) is literally a 2-level IVF search of the centroid
// table: coarse = meta-centroids (top-3, stable order), lists = centroids grouped by
// meta-centroid, k = 1 — so it runs on the same scan/select kernels as search, in LANES order.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <memory>
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

// per-cluster mean with the sum taken in ascending member order (kmeans.rs:693-703).
// one thread per (cluster, dim); keep_old: leave the row untouched when the cluster is empty
// (build_centroid_hierarchy, kmeans.rs:634-638) instead of writing zeros.
__global__ void segment_mean_kernel(const float *X, const uint32_t *order, const uint32_t *seg_off, uint32_t k,
                                    uint32_t d, float *C, int keep_old) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (uint64_t)k * d) return;
  const uint32_t c = (uint32_t)(t / d), j = (uint32_t)(t % d);
  const uint32_t b = seg_off[c], e = seg_off[c + 1];
  float sum = 0.0f;
  for (uint32_t i = b; i < e; ++i) sum += X[(size_t)order[i] * d + j];
  if (e > b) C[t] = sum / (float)(e - b);
  else if (!keep_old) C[t] = 0.0f;
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
  for (int iter = 0; iter < 5; ++iter) {
    VI_TRY(assign_brute_device(cx, Cd, k, meta.p, meta_k, d, d_c2m.p, nullptr, bws));
    VI_HIP(hipMemcpyAsync(c2m.data(), d_c2m.p, k * 4, hipMemcpyDeviceToHost, cx.st));
    VI_HIP(hipStreamSynchronize(cx.st));
    group_by_label(c2m.data(), k, meta_k, order, seg);
    VI_TRY(to_device(d_order, order.data(), order.size(), cx.st));
    VI_TRY(to_device(d_seg, seg.data(), seg.size(), cx.st));
    const uint64_t nt = meta_k * d;
    hipLaunchKernelGGL(segment_mean_kernel, dim3((uint32_t)((nt + 255) / 256)), dim3(256), 0, cx.st, Cd, d_order.p,
                       d_seg.p, (uint32_t)meta_k, d, meta.p, 1);
    VI_HIP(hipGetLastError());
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
  std::vector<float> h_min(m, INFINITY), w(m), cum(m);
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
    float total = 0.0f;
    for (uint64_t j = 0; j < m; ++j) { w[j] = pinned[j] * pinned[j]; total += w[j]; }  // :190,193 (dist^4, sequential)
    if (total == 0.0f) csel[i] = csel[rng.gen_range(0, i)];
    else csel[i] = (uint32_t)rng.weighted_index(w.data(), m, cum.data()) + 1u;
  }
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
// sums[c, j] = sum of X[i, j] over this rank's members of cluster c in ascending i (the reference's order inside
// the rank), counts[c] = members.  One thread per (cluster, dim) over the stable grouping of the rank's points.
__global__ void segment_sum_kernel(const float *X, const uint32_t *order, const uint32_t *seg_off, uint32_t k,
                                   uint32_t d, float *sums, uint32_t *counts) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (uint64_t)k * d) return;
  const uint32_t c = (uint32_t)(t / d), j = (uint32_t)(t % d);
  const uint32_t b = seg_off[c], e = seg_off[c + 1];
  float sum = 0.0f;
  for (uint32_t i = b; i < e; ++i) sum += X[(size_t)order[i] * d + j];
  sums[t] = sum;
  if (j == 0) counts[c] = e - b;
}

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
  DevBuf<uint32_t> d_order, d_seg;
  VI_TRY(Cn.reserve(k * d));
  VI_HIP(hipMemsetAsync(lab, 0, n * 4, cx.st));
  VI_TRY(kmeans_pp_init_rows(cx, src, n, d, k, seed, Cd));
  BruteWs bws;
  std::vector<uint32_t> l32(n), order, seg;
  std::vector<uint64_t> counts(k);
  std::vector<float> h_local;
  uint64_t it = 0;
  for (; it < max_iters; ++it) {
    VI_TRY(assign_device(cx, Xd, n, Cd, k, d, seed, mode, lab, bws));
    VI_HIP(hipMemcpyAsync(l32.data(), lab, n * 4, hipMemcpyDeviceToHost, cx.st));
    VI_HIP(hipStreamSynchronize(cx.st));
    group_by_label(l32.data(), n, k, order, seg);
    VI_TRY(to_device(d_order, order.data(), order.size(), cx.st));
    VI_TRY(to_device(d_seg, seg.data(), seg.size(), cx.st));
    const uint64_t nt = k * d;
    hipLaunchKernelGGL(segment_mean_kernel, dim3((uint32_t)((nt + 255) / 256)), dim3(256), 0, cx.st, Xd, d_order.p,
                       d_seg.p, (uint32_t)k, d, Cn.p, 0);
    VI_HIP(hipGetLastError());
    VI_HIP(hipStreamSynchronize(cx.st));
    for (uint64_t c = 0; c < k; ++c) counts[c] = seg[c + 1] - seg[c];
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
  std::vector<uint32_t> perm(n), bidx(B), blab(B), members, tc, ts, tl;
  std::vector<float> teta, h_local;
  std::vector<uint32_t> head(k), nxt(B);
  BruteWs bws;
  uint64_t it = 0;
  for (; it < max_iters; ++it) {
    // sample_batch (kmeans.rs:722-726): full shuffle of 0..n, first B
    for (uint64_t i = 0; i < n; ++i) perm[i] = (uint32_t)i;
    rng.shuffle(perm.data(), n);
    std::copy(perm.begin(), perm.begin() + B, bidx.begin());
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
  DevBuf<uint32_t> d_order, d_seg, d_bad;
  std::vector<uint64_t> off;
  VI_TRY(d_bad.reserve(1));
  VI_HIP(hipMemsetAsync(d_bad.p, 0, 4, cx.st));
  if (n) {
    hipLaunchKernelGGL(label_range_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, cx.st, labels_dev, n, (uint32_t)k, d_bad.p);
    VI_HIP(hipGetLastError());
  }
  uint32_t bad = 0;
  VI_HIP(hipMemcpyAsync(&bad, d_bad.p, 4, hipMemcpyDeviceToHost, cx.st));
  VI_HIP(hipStreamSynchronize(cx.st));
  if (bad) return fail(VI_ERR_INVALID_INPUT, "a label is not below k");
  VI_TRY(group_ids_by_label_device(labels_dev, n, k, d_order, off, cx.st));
  std::vector<uint32_t> seg(off.begin(), off.end());
  VI_TRY(to_device(d_seg, seg.data(), seg.size(), cx.st));
  const uint64_t nt = k * d;
  hipLaunchKernelGGL(segment_sum_kernel, dim3((uint32_t)((nt + 255) / 256)), dim3(256), 0, cx.st, Xd, d_order.p, d_seg.p,
                     (uint32_t)k, d, sums_dev, counts_dev);
  VI_HIP(hipGetLastError());
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
