// search_kernels.hip — gfx950 kernels of the IVF search path and their host-side pipeline.
//
// Reference path being replaced: IvfIndex::search_with_paths (src/ivf_index.rs:190-267):
//   coarse  : euclidean_distance_squared(q, c_i) for all centroids, stable sort, take n_probe
//   scan    : euclidean_distance_squared(q, v) for every vector of the probed lists
//   select  : stable sort of all candidates, take k
// with euclidean_distance_squared = strictly sequential f32 sum of (x-y)^2 (src/utils.rs:28-30).
//
// GPU formulation (all distances are computed in the reference's exact summation order, so
// ids AND distances are bit-identical; nothing is re-ranked):
//   scan_kernel<COARSE>   one wave = (group of QG queries) x (range of centroid blocks);
//                         lane = centroid, per-lane sequential f32 chain over d; wave-resident
//                         sorted top-P list per query (DPP shift insertion)
//   coarse_merge_kernel   one wave per query: S sorted partial runs -> probe list, shard
//                         visiting order, histogram of probed lists
//   group_*_kernel        counting sort of (query,probe) pairs by list  => every list block is
//                         streamed once per group of QG queries that probe it
//   scan_kernel<LISTS>    one wave = (list) x (group of <= QG queries probing it)
//   final_merge_kernel    one wave per query: P sorted runs -> top-k, ids, tie keys
//
// Compiled with -ffp-contract=off: a fused multiply-add would change the rounding of
// acc + t*t and break bit parity with the reference.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <memory>

#include "device_index.hpp"
#include "device_math.hpp"
#include "scan.hpp"
#include "wave_select.hpp"
#include "wave_sort.hpp"

namespace vi {
namespace {

constexpr int kWave = 64;
constexpr int kWavesPerBlock = 4;
constexpr int kBlockThreads = kWave * kWavesPerBlock;

// Minimum blocks (x64 vectors) per list segment.  Measured on the C2 workload (profiles/
// r01_experiments.md): cutting lists finer than this costs more in repeated top-k warm-up than it
// gains in load balance, so only extreme lists (> 64k vectors) are cut (seg 256 measured +9 %).
constexpr uint32_t kSegBlocksDefault = 1024;

// tuning knobs for experiments (scripts/gpu_scan_bench.py); unset => defaults
inline uint32_t env_u32(const char *name, uint32_t dflt) {
  const char *v = getenv(name);
  return (v && *v) ? (uint32_t)strtoul(v, nullptr, 10) : dflt;
}

// ------------------------------------------------------------------------------------------
// Exact-order accumulators.  SCALAR: src/utils.rs:28-30.  LANES: src/kmeans.rs:377-419
// (8-lane chunks, then one 4-lane chunk, then a scalar tail; reduce order see oracle header).
// ------------------------------------------------------------------------------------------
// Accumulator state of QG (query, vector) pairs per lane.  add<T>() consumes quad T (0..3) of the
// current group of 4 quads; `qi` is the absolute quad index (wave-uniform).
template <int ORDER, int QG>
struct Acc;

template <int QG>
struct Acc<VI_ORDER_SCALAR, QG> {
  float a[QG];
  __device__ __forceinline__ void reset() {
#pragma unroll
    for (int j = 0; j < QG; ++j) a[j] = 0.0f;
  }
  template <int T>
  __device__ __forceinline__ void add(int j, uint32_t, uint32_t, bool, const float4 &q, const float4 &x) {
    sq_add(a[j], q.x, x.x);
    sq_add(a[j], q.y, x.y);
    sq_add(a[j], q.z, x.z);
    sq_add(a[j], q.w, x.w);
  }
  __device__ __forceinline__ float finish(int j) const { return a[j]; }
};

template <int QG>
struct Acc<VI_ORDER_LANES, QG> {
  float a8[QG][8], a4[QG][4], tail[QG];
  __device__ __forceinline__ void reset() {
#pragma unroll
    for (int j = 0; j < QG; ++j) {
#pragma unroll
      for (int l = 0; l < 8; ++l) a8[j][l] = 0.0f;
      a4[j][0] = a4[j][1] = a4[j][2] = a4[j][3] = 0.0f;
      tail[j] = 0.0f;
    }
  }
  // n8x2 = 2*(dim/8): quads below it feed the 8 lane accumulators (kmeans.rs:387-396); the next
  // quad is the f32x4 chunk if at least 4 dims remain (:399-408); what is left is the scalar
  // tail (:411-416).  Zero padding contributes exact +0 wherever it lands.
  template <int T>
  __device__ __forceinline__ void add(int j, uint32_t qi, uint32_t n8x2, bool has4, const float4 &q,
                                      const float4 &x) {
    if (qi < n8x2) {
      constexpr int h = (T & 1) * 4;
      sq_add(a8[j][h + 0], q.x, x.x); sq_add(a8[j][h + 1], q.y, x.y);
      sq_add(a8[j][h + 2], q.z, x.z); sq_add(a8[j][h + 3], q.w, x.w);
    } else if (qi == n8x2 && has4) {
      sq_add(a4[j][0], q.x, x.x); sq_add(a4[j][1], q.y, x.y);
      sq_add(a4[j][2], q.z, x.z); sq_add(a4[j][3], q.w, x.w);
    } else {
      sq_add(tail[j], q.x, x.x); sq_add(tail[j], q.y, x.y);
      sq_add(tail[j], q.z, x.z); sq_add(tail[j], q.w, x.w);
    }
  }
  __device__ __forceinline__ float finish(int j) const {
    const float lo = VI_REDUCE4(a8[j][0], a8[j][1], a8[j][2], a8[j][3]);  // include/vi_reduce_order.h
    const float hi = VI_REDUCE4(a8[j][4], a8[j][5], a8[j][6], a8[j][7]);
    const float r4 = VI_REDUCE4(a4[j][0], a4[j][1], a4[j][2], a4[j][3]);
    return ((lo + hi) + r4) + tail[j];
  }
};

// ------------------------------------------------------------------------------------------
// scan kernel
// ------------------------------------------------------------------------------------------
template <int QG, int ORDER, bool COARSE, bool DUMP>
__global__ void __launch_bounds__(kBlockThreads) scan_kernel(ScanArgs a) {
  extern __shared__ float4 smem[];
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x >> 6;
  float4 *lq = smem + (size_t)wave * QG * a.dq;
  const uint32_t item = blockIdx.x * kWavesPerBlock + wave;

  uint32_t nqi, b0, b1, len, fb;
  uint32_t nseg = 1, seg = 0, segrun0 = 0;
  uint32_t slot[QG], qid[QG];
  if (COARSE) {
    const uint32_t nqg = (a.nq + QG - 1) / QG;
    if (item >= nqg * a.S) return;
    const uint32_t qc = item / a.S, sp = item - qc * a.S;
    const uint32_t q0 = qc * QG;
    nqi = min((uint32_t)QG, a.nq - q0);
#pragma unroll
    for (int j = 0; j < QG; ++j) { qid[j] = q0 + j; slot[j] = (q0 + j) * a.S + sp; }
    const uint32_t nblk = (a.nvec + kWave - 1) / kWave;
    b0 = sp * a.bps;
    b1 = min(nblk, b0 + a.bps);
    fb = 0;
    len = a.nvec;
  } else {
    const uint32_t nitems = a.item_start[a.nlists];
    if (item >= nitems) return;
    // largest l with item_start[l] <= item  (lists with no items have equal neighbours)
    uint32_t lo = 0, hi = a.nlists;
    while (hi - lo > 1) {
      const uint32_t mid = (lo + hi) >> 1;
      if (a.item_start[mid] <= item) lo = mid; else hi = mid;
    }
    const uint32_t l = (uint32_t)__builtin_amdgcn_readfirstlane((int)lo);
    const uint32_t s0 = a.seg_start[l], cnt = a.seg_start[l + 1] - s0;
    len = a.list_len[l];
    uint32_t segb;
    nseg = list_segments(len, a.segb0, &segb);
    const uint32_t local = item - a.item_start[l];
    const uint32_t chunk = local / nseg;
    seg = local - chunk * nseg;
    const uint32_t j0 = chunk * QG;
    nqi = min((uint32_t)QG, cnt - j0);
#pragma unroll
    for (int j = 0; j < QG; ++j) {
      const uint32_t s = (j < (int)nqi) ? a.pairs[s0 + j0 + j] : 0u;
      slot[j] = (uint32_t)__builtin_amdgcn_readfirstlane((int)s);
      qid[j] = slot[j] / a.P;
    }
    segrun0 = nseg > 1 ? a.segrun_start[l] + j0 * nseg + seg : 0u;
    fb = a.first_block[l];
    const uint32_t nblk = (len + kWave - 1) / kWave;
    b0 = seg * segb;
    b1 = min(nblk, b0 + segb);
    if (a.max_blocks) b1 = min(b1, a.max_blocks);
  }

  // stage the group's queries in this wave's LDS slice, zero padded to dq*4 floats
  {
    float *lqf = reinterpret_cast<float *>(lq);
    const uint32_t dpad = a.dq * 4;
#pragma unroll
    for (int j = 0; j < QG; ++j) {
      const float *src = a.Q + (size_t)qid[j] * a.dim;
      const bool live = j < (int)nqi;
      for (uint32_t e = lane; e < dpad; e += kWave) lqf[j * dpad + e] = (live && e < a.dim) ? src[e] : 0.0f;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }

  WaveTopK sel[QG];
#pragma unroll
  for (int j = 0; j < QG; ++j) sel[j].init();
  const int K = (int)a.K;

  // Flat software-pipelined sweep over (block, group-of-4-quads): the next group's four
  // global_load_dwordx4 are issued before the current group is consumed, also across block
  // boundaries, so the selection step of a block overlaps the first loads of the next one.
  const uint32_t gpb = a.dq >> 2;  // dq is a multiple of 4 (layout pads dims to 16)
  const uint32_t total = (b1 - b0) * gpb;
  const uint32_t n8x2 = 2 * (a.dim / 8);
  const bool has4 = (a.dim - 4 * n8x2) >= 4;
  const float4 *vbase = a.blocks + ((size_t)(fb + b0) * a.dq) * kWave + lane;
  Acc<ORDER, QG> acc;
  acc.reset();
  float4 xn0, xn1, xn2, xn3;
  if (total) {
    xn0 = vbase[0]; xn1 = vbase[kWave]; xn2 = vbase[2 * kWave]; xn3 = vbase[3 * kWave];
  }
  uint32_t gq = 0, blk = b0;
  for (uint32_t g = 0; g < total; ++g) {
    const float4 x0 = xn0, x1 = xn1, x2 = xn2, x3 = xn3;
    if (g + 1 < total) {
      const float4 *nx = vbase + (size_t)(g + 1) * 4 * kWave;  // groups are contiguous in memory
      xn0 = nx[0]; xn1 = nx[kWave]; xn2 = nx[2 * kWave]; xn3 = nx[3 * kWave];
    }
    const uint32_t qi = gq * 4;
#pragma unroll
    for (int j = 0; j < QG; ++j) acc.template add<0>(j, qi + 0, n8x2, has4, lq[j * a.dq + qi + 0], x0);
#pragma unroll
    for (int j = 0; j < QG; ++j) acc.template add<1>(j, qi + 1, n8x2, has4, lq[j * a.dq + qi + 1], x1);
#pragma unroll
    for (int j = 0; j < QG; ++j) acc.template add<2>(j, qi + 2, n8x2, has4, lq[j * a.dq + qi + 2], x2);
#pragma unroll
    for (int j = 0; j < QG; ++j) acc.template add<3>(j, qi + 3, n8x2, has4, lq[j * a.dq + qi + 3], x3);
    if (++gq == gpb) {
      const uint32_t pos = blk * kWave + lane;
      const bool valid = pos < len;
      const uint32_t p = valid ? pos : kNoPos;
#pragma unroll
      for (int j = 0; j < QG; ++j)
        if (j < (int)nqi) {
          const float dj = valid ? acc.finish(j) : INFINITY;
          if (DUMP) {  // generic path: every candidate is written out, nothing is selected
            if (valid) {
              const uint32_t idx = COARSE ? pos : a.dump_off[slot[j]] + pos;
              const uint64_t row = COARSE ? (uint64_t)qid[j] : (uint64_t)(slot[j] / a.P);
              a.dump_keys[row * a.dump_row + idx] = ((uint64_t)__float_as_uint(dj) << 32) | idx;
            }
          } else {
            sel[j].offer(dj, p, K);
          }
        }
      acc.reset();
      gq = 0;
      ++blk;
    }
  }

  if (!DUMP && lane < K) {
#pragma unroll
    for (int j = 0; j < QG; ++j)
      if (j < (int)nqi) {
        if (COARSE || nseg == 1) {
          a.run_dist[(size_t)slot[j] * K + lane] = sel[j].d;
          a.run_pos[(size_t)slot[j] * K + lane] = sel[j].p;
        } else {
          const size_t r = (size_t)(segrun0 + (uint32_t)j * nseg) * K + lane;
          a.seg_run_dist[r] = sel[j].d;
          a.seg_run_pos[r] = sel[j].p;
        }
      }
  }
}

// merge the segment runs of every (query, list) pair whose list was cut into segments
struct SegMergeArgs {
  const uint32_t *seg_start, *segrun_start, *list_len, *pairs;
  uint32_t nlists, segb0, K;
  const float *seg_run_dist;
  const uint32_t *seg_run_pos;
  float *run_dist;
  uint32_t *run_pos;
};

__global__ void __launch_bounds__(kBlockThreads) seg_merge_kernel(SegMergeArgs a) {
  const int lane = threadIdx.x & (kWave - 1);
  const uint32_t pi = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  if (pi >= a.seg_start[a.nlists]) return;
  uint32_t lo = 0, hi = a.nlists;
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (a.seg_start[mid] <= pi) lo = mid; else hi = mid;
  }
  const uint32_t l = (uint32_t)__builtin_amdgcn_readfirstlane((int)lo);
  uint32_t segb;
  const uint32_t nseg = list_segments(a.list_len[l], a.segb0, &segb);
  if (nseg <= 1) return;
  const uint32_t K = a.K;
  const size_t base = ((size_t)a.segrun_start[l] + (size_t)(pi - a.seg_start[l]) * nseg + lane) * K;
  uint32_t head = 0, hp = kNoPos;
  float hd = INFINITY;
  if ((uint32_t)lane < nseg) { hd = a.seg_run_dist[base]; hp = a.seg_run_pos[base]; }
  float od = INFINITY;
  uint32_t op = kNoPos;
  for (uint32_t i = 0; i < K; ++i) {
    // segments hold increasing positions, so (dist, segment) == (dist, position) order
    const uint64_t key = (hp == kNoPos) ? ~0ull : (((uint64_t)__float_as_uint(hd) << 32) | (uint32_t)lane);
    const uint64_t m = wave_min_u64(key);
    if (m == ~0ull) break;
    const int win = (int)(uint32_t)m;
    const uint32_t wp = readlane_u(hp, win);
    if ((uint32_t)lane == i) { od = __uint_as_float((uint32_t)(m >> 32)); op = wp; }
    if (lane == win) {
      ++head;
      if (head < K) { hd = a.seg_run_dist[base + head]; hp = a.seg_run_pos[base + head]; }
      else hp = kNoPos;
    }
  }
  if ((uint32_t)lane < K) {
    const size_t o = (size_t)a.pairs[pi] * K + lane;
    a.run_dist[o] = od;
    a.run_pos[o] = op;
  }
}

// ------------------------------------------------------------------------------------------
// coarse merge: S sorted runs per query -> probes (rank order), shard visiting order, histogram
// ------------------------------------------------------------------------------------------
struct CoarseMergeArgs {
  const float *run_dist;
  const uint32_t *run_pos;
  uint32_t nq, S, P;              // P = entries per run = probes per query (<= 64), S <= 64
  const uint32_t *list_shard, *list_len;
  uint32_t *probes, *gorder, *cnt;
  uint32_t nlists;
};

__global__ void __launch_bounds__(kBlockThreads) coarse_merge_kernel(CoarseMergeArgs a) {
  const int lane = threadIdx.x & (kWave - 1);
  const uint32_t q = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  if (q >= a.nq) return;
  const size_t base = ((size_t)q * a.S + lane) * a.P;
  uint32_t head = 0;
  float hd = INFINITY;
  uint32_t hp = kNoPos;
  if (lane < (int)a.S) { hd = a.run_dist[base]; hp = a.run_pos[base]; }
  uint32_t mylist = kNoPos;
  uint32_t found = 0;
  for (uint32_t i = 0; i < a.P; ++i) {
    // key = (dist, centroid index): stable sort over index order (ivf_index.rs:205-215)
    const uint64_t key = (hp == kNoPos) ? ~0ull : (((uint64_t)__float_as_uint(hd) << 32) | hp);
    const uint64_t m = wave_min_u64(key);
    if (m == ~0ull) break;
    if ((uint32_t)lane == i) mylist = (uint32_t)m;
    if (key == m) {
      ++head;
      if (head < a.P) { hd = a.run_dist[base + head]; hp = a.run_pos[base + head]; }
      else hp = kNoPos;
    }
    ++found;
  }
  const bool live = (uint32_t)lane < found;
  const uint32_t g = probe_candidate_order(lane, found, mylist, a.list_shard);
  if ((uint32_t)lane < a.P) {
    a.probes[(size_t)q * a.P + lane] = live ? mylist : kNoPos;
    a.gorder[(size_t)q * a.P + lane] = live ? g : kNoPos;
    if (live && a.list_len[mylist] > 0) atomicAdd(&a.cnt[subbin_index(mylist, q & (kSubBins - 1), a.nlists)], 1u);
  }
}

// ------------------------------------------------------------------------------------------
// grouping (counting sort of (query, probe) pairs by list)
// ------------------------------------------------------------------------------------------
// single block (cnt = the per-list totals of list_totals_kernel): exclusive scans over the lists of
//   seg_start    Σ cnt                      (pairs grouped by list)
//   item_start   Σ ceil(cnt/QG) * nseg      (scan work items)
//   segrun_start Σ cnt * nseg [nseg > 1]    (segment runs awaiting seg_merge_kernel)
// stats[0] = Σ cnt*len, stats[1] = items, stats[2] = segment runs
// queries probing every list: the sum of its sub-bin counters (one thread per list: coalesced along each sub-bin row)
// (also resets the counters group_scan_kernel adds to — stats[0..5] and [12] — when `stats` is given: two memset
// launches less on a path made of 5-microsecond kernels)
__global__ void list_totals_kernel(const uint32_t *cnt, uint32_t nlists, uint32_t *tot, uint64_t *stats) {
  const uint32_t l = blockIdx.x * blockDim.x + threadIdx.x;
  if (stats && l < 7) stats[l < 6 ? l : 12] = 0;
  if (l >= nlists) return;
  const uint32_t st = subbin_stride(nlists);
  uint32_t c = 0;
#pragma unroll
  for (uint32_t s = 0; s < kSubBins; ++s) c += cnt[s * st + l];
  tot[l] = c;
}

// where each sub-bin of a list scatters to: its own slice of the list's segment of `pairs`
__global__ void cursor_kernel(const uint32_t *cnt, const uint32_t *seg_start, uint32_t nlists, uint32_t *cursor) {
  const uint32_t l = blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= nlists) return;
  const uint32_t st = subbin_stride(nlists);
  uint32_t run = seg_start[l];
#pragma unroll
  for (uint32_t s = 0; s < kSubBins; ++s) {
    cursor[s * st + l] = run;
    run += cnt[s * st + l];
  }
}

// the scan over the lists shared by group_scan_kernel and group_prepare_kernel (one workgroup of 1024 threads)
__device__ __forceinline__ void group_scan_lists(const uint32_t *cnt, const uint32_t *list_len, uint32_t nlists, uint32_t qg,
                                                 uint32_t segb0, uint32_t *seg_start, uint32_t *item_start, uint32_t *segrun_start,
                                                 uint64_t *stats, uint32_t *tile_start, uint32_t *s_seg, uint32_t *s_item,
                                                 uint32_t *s_run, uint32_t *s_tile) {
  const uint32_t t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  // ---- lists (cnt = the per-list totals of list_totals_kernel, which also reset the counters added to below) ----
  // Wave w owns the contiguous lists [w R 64, (w + 1) R 64) as R rows of 64, a lane per list: every load and store is one
  // coalesced instruction, eight rows' loads in flight together.  (A thread walking its own 64 lists — 65 536 lists — read
  // and wrote with a stride of 256 bytes between lanes, 64 cache lines per instruction, all from the one CU this scan runs
  // on: 0.34 ms of a 4.9 ms search.)  Pass 1: the wave's totals; pass 2, behind the workgroup's prefix over the waves: a
  // DPP scan per row and quantity, the carry from row to row.
  const uint32_t rows = ((nlists + 63u) / 64u + 15u) / 16u;  // rows of 64 lists per wave
  const uint32_t l_base = (uint32_t)wave * rows * 64u;
  struct PerList { uint32_t seg, item, run, tile; };
  auto per_list = [&](uint32_t c, uint32_t len, unsigned long long *v4) {
    uint32_t segb;
    const uint32_t ns = list_segments(len, segb0, &segb);
    const uint32_t chunks = group_chunks(c, qg);
    if (v4) {
      v4[0] += (unsigned long long)c * len;
      v4[1] += 2ull * c * ns;
      v4[2] += (unsigned long long)chunks * ((len + 63) / 64);
      v4[3] += (unsigned long long)((c + 127) / 128) * ((len + 63) / 64);
    }
    return PerList{c, chunks * ns, ns > 1 ? c * ns : 0u, chunks * ns * seg_records(segb)};
  };
  uint32_t seg = 0, item = 0, run = 0, tile = 0;  // this lane's column sums over the wave's rows
  unsigned long long v4[4] = {0, 0, 0, 0};        // vec, rec, mtile, mtile128
  for (uint32_t r0 = 0; r0 < rows; r0 += 8) {
    uint32_t cs[8], lens[8];
#pragma unroll
    for (uint32_t u = 0; u < 8; ++u) {
      const uint32_t l = l_base + (r0 + u) * 64u + (uint32_t)lane;
      const bool in = r0 + u < rows && l < nlists;
      cs[u] = in ? cnt[l] : 0u;
      lens[u] = in ? list_len[l] : 0u;
    }
#pragma unroll
    for (uint32_t u = 0; u < 8; ++u) {
      const PerList p = per_list(cs[u], lens[u], v4);  // (rows past the end: zeros)
      seg += p.seg; item += p.item; run += p.run; tile += p.tile;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    seg += (uint32_t)__shfl_xor((int)seg, o); item += (uint32_t)__shfl_xor((int)item, o);
    run += (uint32_t)__shfl_xor((int)run, o); tile += (uint32_t)__shfl_xor((int)tile, o);
    v4[0] += __shfl_xor(v4[0], o); v4[1] += __shfl_xor(v4[1], o); v4[2] += __shfl_xor(v4[2], o); v4[3] += __shfl_xor(v4[3], o);
  }
  if (lane == 0) {
    s_seg[wave] = seg; s_item[wave] = item; s_run[wave] = run; s_tile[wave] = tile;
    atomicAdd((unsigned long long *)&stats[0], v4[0]);
    atomicAdd((unsigned long long *)&stats[4], v4[1]);
    atomicAdd((unsigned long long *)&stats[3], v4[2]);
    atomicAdd((unsigned long long *)&stats[12], v4[3]);
  }
  __syncthreads();
  uint32_t rs = 0, ri = 0, rr = 0, rt = 0, tseg = 0, titem = 0, trun = 0, ttile = 0;
  for (int w = 0; w < 16; ++w) {
    if (w < wave) { rs += s_seg[w]; ri += s_item[w]; rr += s_run[w]; rt += s_tile[w]; }
    tseg += s_seg[w]; titem += s_item[w]; trun += s_run[w]; ttile += s_tile[w];
  }
  for (uint32_t r0 = 0; r0 < rows; r0 += 8) {  // (the same values again, from L2 now)
    uint32_t cs[8], lens[8];
#pragma unroll
    for (uint32_t u = 0; u < 8; ++u) {
      const uint32_t l = l_base + (r0 + u) * 64u + (uint32_t)lane;
      const bool in = r0 + u < rows && l < nlists;
      cs[u] = in ? cnt[l] : 0u;
      lens[u] = in ? list_len[l] : 0u;
    }
#pragma unroll
    for (uint32_t u = 0; u < 8; ++u) {
      if (r0 + u >= rows) break;  // (wave-uniform)
      const uint32_t l = l_base + (r0 + u) * 64u + (uint32_t)lane;
      const PerList p = per_list(cs[u], lens[u], nullptr);
      const uint32_t is = wave_incl_scan_u32(p.seg), ii = wave_incl_scan_u32(p.item);
      const uint32_t ir = wave_incl_scan_u32(p.run), it = wave_incl_scan_u32(p.tile);
      if (l < nlists) {
        seg_start[l] = rs + is - p.seg; item_start[l] = ri + ii - p.item; segrun_start[l] = rr + ir - p.run;
        if (tile_start) tile_start[l] = rt + it - p.tile;
      }
      rs += readlane_u(is, 63); ri += readlane_u(ii, 63); rr += readlane_u(ir, 63); rt += readlane_u(it, 63);
    }
  }
  if (t == 0) {
    seg_start[nlists] = tseg;
    item_start[nlists] = titem;
    segrun_start[nlists] = trun;
    stats[1] = titem;
    stats[2] = trun;
    stats[5] = ttile;
  }
}

__global__ void __launch_bounds__(1024) group_scan_kernel(const uint32_t *cnt, const uint32_t *list_len,
                                                          uint32_t nlists, uint32_t qg, uint32_t segb0,
                                                          uint32_t *seg_start, uint32_t *item_start,
                                                          uint32_t *segrun_start, uint64_t *stats,
                                                          uint32_t *tile_start) {
  __shared__ uint32_t s_seg[16], s_item[16], s_run[16], s_tile[16];
  group_scan_lists(cnt, list_len, nlists, qg, segb0, seg_start, item_start, segrun_start, stats, tile_start, s_seg, s_item, s_run, s_tile);
}

// group_scan_kernel (workgroup 0) and the queries' record offsets (workgroup 1: exclusive scan of qtot, qoff[nq] = total)
// in one launch: both are single-workgroup scans, independent of each other.  (Folding list_totals and cursor in as well —
// one workgroup reading all 32 sub-bins of every list twice — was measured: the grouping took twice as long.)
__global__ void __launch_bounds__(1024) group_prepare_kernel(const uint32_t *cnt, const uint32_t *list_len, uint32_t nlists, uint32_t qg,
                                                             uint32_t segb0, uint32_t *seg_start, uint32_t *item_start,
                                                             uint32_t *segrun_start, uint64_t *stats, uint32_t *tile_start,
                                                             const uint32_t *qtot, uint32_t nq, uint32_t *qoff) {
  __shared__ uint32_t s_seg[16], s_item[16], s_run[16], s_tile[16];
  const uint32_t t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  if (blockIdx.x == 1) {  // ---- query offsets ----
    if (!qtot) return;
    const uint32_t per = (nq + 1023) / 1024;
    const uint32_t beg = min(nq, t * per), end = min(nq, beg + per);
    uint32_t sum = 0;
    for (uint32_t i = beg; i < end; ++i) sum += qtot[i];
    uint32_t inc = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t x = (uint32_t)__shfl_up((int)inc, o);
      if (lane >= o) inc += x;
    }
    if (lane == 63) s_seg[wave] = inc;
    __syncthreads();
    uint32_t w = 0, tot = 0;
    for (int i = 0; i < 16; ++i) {
      if (i < wave) w += s_seg[i];
      tot += s_seg[i];
    }
    uint32_t run = w + inc - sum;
    for (uint32_t i = beg; i < end; ++i) { qoff[i] = run; run += qtot[i]; }
    if (t == 0) qoff[nq] = tot;
    return;
  }
  group_scan_lists(cnt, list_len, nlists, qg, segb0, seg_start, item_start, segrun_start, stats, tile_start, s_seg, s_item, s_run, s_tile);
}

// (list ids are range-checked wherever they index: a caller-supplied probe list, vi_indexer_search_probed_device, is
// validated up front by validate_probes_kernel, and a stray word can then still not fault the GPU)
__global__ void histogram_kernel(const uint32_t *probes, const uint32_t *list_len, uint32_t nlists, uint32_t n, uint32_t P,
                                 uint32_t *cnt) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t l = probes[i];
  if (l < nlists && list_len[l] > 0) atomicAdd(&cnt[subbin_index(l, div_probes(i, P) & (kSubBins - 1), nlists)], 1u);
}

__global__ void group_scatter_kernel(const uint32_t *probes, const uint32_t *list_len, uint32_t nlists, uint32_t P,
                                     uint32_t *cursor, uint32_t *pairs, uint32_t total, const uint32_t *seg_start,
                                     uint32_t *pair_pos) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const uint32_t l = probes[i];
  if (l >= nlists || list_len[l] == 0) return;
  const uint32_t pos = atomicAdd(&cursor[subbin_index(l, div_probes(i, P) & (kSubBins - 1), nlists)], 1u);
  pairs[pos] = i;  // slot id = q*P + rank
  if (pair_pos) pair_pos[i] = pos - seg_start[l];  // MFMA path: where the pair sits among the pairs of its list
}

// the same without atomics: the pair's place among the pairs of its (list, sub-bin) came back from the histogram
// increment of the kernel that chose the probe (coarse_select_direct_kernel) — 320 000 returning atomics on counters
// shared across the XCDs were most of the scatter's 17 us
__global__ void group_scatter_ranked_kernel(const uint32_t *probes, const uint32_t *list_len, uint32_t nlists, uint32_t P,
                                            const uint32_t *cursor, const uint32_t *pair_rank, uint32_t *pairs, uint32_t total,
                                            const uint32_t *seg_start, uint32_t *pair_pos) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const uint32_t l = probes[i];
  if (l >= nlists || list_len[l] == 0) return;
  const uint32_t pos = cursor[subbin_index(l, div_probes(i, P) & (kSubBins - 1), nlists)] + pair_rank[i];
  pairs[pos] = i;
  if (pair_pos) pair_pos[i] = pos - seg_start[l];
}

// caller-supplied probe lists (multi-GPU: another rank's coarse step): every probe must name a list of this index or be the
// empty marker, the real probes of a row must come first, and the candidate-order ranks must be a rank < P or the marker
__global__ void validate_probes_kernel(const uint32_t *probes, const uint32_t *order, uint32_t total, uint32_t P,
                                       uint32_t nlists, uint32_t *bad) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const uint32_t l = probes[i], g = order[i];
  bool ok = (l < nlists && g < P) || (l == kNoPos);
  if (ok && l != kNoPos && (i % P) != 0 && probes[i - 1] == kNoPos) ok = false;
  if (!ok) atomicOr(bad, 1u);
}

// ------------------------------------------------------------------------------------------
// final merge: P sorted runs per query -> top-k in the reference's stable candidate order
// ------------------------------------------------------------------------------------------
struct FinalMergeArgs {
  const float *run_dist;
  const uint32_t *run_pos;
  uint32_t nq, P, K, k;           // K entries per run; k outputs per query
  const uint32_t *probes, *gorder, *first_block;
  const uint64_t *ext_ids;
  float *D;
  int64_t *I;
  uint64_t *tie;                  // optional
  uint64_t *slots;                // optional
  uint32_t *counts;               // optional
};

__global__ void __launch_bounds__(kBlockThreads) final_merge_kernel(FinalMergeArgs a) {
  const int lane = threadIdx.x & (kWave - 1);
  const uint32_t q = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  if (q >= a.nq) return;
  const size_t slot = (size_t)q * a.P + lane;
  const size_t base = slot * a.K;
  uint32_t head = 0, hp = kNoPos, g = kNoPos, list = kNoPos;
  float hd = INFINITY;
  if (lane < (int)a.P) {
    list = a.probes[slot];
    g = a.gorder[slot];
    if (list != kNoPos) { hd = a.run_dist[base]; hp = a.run_pos[base]; }
  }
  uint32_t found = 0;
  const uint32_t kk = a.k < a.P * a.K ? a.k : a.P * a.K;
  for (uint32_t i = 0; i < kk; ++i) {
    // (dist, shard-visit/probe order g, position): the stable sort of ivf_index.rs:265
    const uint64_t key = (hp == kNoPos) ? ~0ull : (((uint64_t)__float_as_uint(hd) << 32) | g);
    const uint64_t m = wave_min_u64(key);
    if (m == ~0ull) break;
    if (key == m) {
      const size_t o = (size_t)q * a.k + i;
      const uint64_t gslot = (uint64_t)a.first_block[list] * kWave + hp;
      a.D[o] = hd;
      a.I[o] = (int64_t)a.ext_ids[gslot];
      if (a.tie) a.tie[o] = ((uint64_t)g << 32) | hp;
      if (a.slots) a.slots[o] = gslot;
      ++head;
      if (head < a.K) { hd = a.run_dist[base + head]; hp = a.run_pos[base + head]; }
      else hp = kNoPos;
    }
    ++found;
  }
  for (uint32_t i = found + lane; i < a.k; i += kWave) {  // lib.rs:179-187 padding
    const size_t o = (size_t)q * a.k + i;
    a.D[o] = INFINITY;
    a.I[o] = -1;
    if (a.tie) a.tie[o] = ~0ull;
    if (a.slots) a.slots[o] = ~0ull;
  }
  if (a.counts && lane == 0) a.counts[q] = found;
}

// merge `parts` partial top-k lists per query (multi-GPU): key = (dist, tie)
// part p of Dp / Ip / Tp starts p * stride elements (of the respective type) after part 0
__global__ void __launch_bounds__(kBlockThreads) merge_partials_kernel(uint64_t nq, uint32_t k, uint32_t parts,
                                                                      const float *Dp0, const int64_t *Ip0,
                                                                      const uint64_t *Tp0, size_t strideD,
                                                                      size_t strideI, float *D, int64_t *I) {
  const int lane = threadIdx.x & (kWave - 1);
  const uint64_t q = (uint64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  if (q >= nq) return;
  // lane = part (parts <= 64); each part's list is sorted by (dist, tie) already
  const float *Dp = Dp0 + (size_t)(lane < (int)parts ? lane : 0) * strideD;
  const int64_t *Ip = Ip0 + (size_t)(lane < (int)parts ? lane : 0) * strideI;
  const uint64_t *Tp = Tp0 + (size_t)(lane < (int)parts ? lane : 0) * strideI;
  const size_t base = (size_t)q * k;
  uint32_t head = 0;
  bool live = lane < (int)parts;
  float hd = INFINITY;
  uint64_t ht = ~0ull;
  int64_t hi = -1;
  // an empty slot is marked by its tie key (ids are arbitrary u64: one >= 2^63 reads as a negative i64)
  if (live) { hd = Dp[base]; ht = Tp[base]; hi = Ip[base]; live = ht != ~0ull; }
  uint32_t found = 0;
  for (uint32_t i = 0; i < k; ++i) {
    // two-level key: distance bits first, then the 64-bit tie key
    const uint64_t kd = live ? (uint64_t)__float_as_uint(hd) : ~0ull;
    const uint64_t md = wave_min_u64(kd);
    if (md == ~0ull) break;
    const uint64_t kt = (live && kd == md) ? ht : ~0ull;
    const uint64_t mt = wave_min_u64(kt);
    if (live && kd == md && kt == mt) {
      D[q * k + i] = hd;
      I[q * k + i] = hi;
      ++head;
      if (head < k) { hd = Dp[base + head]; ht = Tp[base + head]; hi = Ip[base + head]; live = ht != ~0ull; }
      else live = false;
    }
    ++found;
  }
  for (uint32_t i = found + lane; i < k; i += kWave) { D[q * k + i] = INFINITY; I[q * k + i] = -1; }
}

// ------------------------------------------------------------------------------------------
// repack: AoS records (or row-major rows) -> lane-interleaved blocks (+ external ids)
// ------------------------------------------------------------------------------------------
struct RepackArgs {
  const uint8_t *src;           // device image of a shard file / row-major matrix
  const uint64_t *blk_src;      // [nb] byte offset of the first record of each destination block
  const uint32_t *blk_nv;       // [nb] valid vectors in the block (1..64)
  const uint32_t *blk_dst;      // [nb] destination block index
  uint32_t nb, dim, dq;
  uint32_t stride;              // bytes between consecutive records
  uint32_t vec_off;             // byte offset of the f32 payload inside a record
  int32_t id_off;               // byte offset of the u64 external id, or -1
  float *blocks;
  uint64_t *ext_ids;            // may be null
};

__global__ void __launch_bounds__(kBlockThreads) repack_kernel(RepackArgs a) {
  const int lane = threadIdx.x & (kWave - 1);
  const uint32_t b = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  if (b >= a.nb) return;
  const uint32_t nv = a.blk_nv[b], dst = a.blk_dst[b];
  const uint8_t *rec = a.src + a.blk_src[b] + (size_t)lane * a.stride;
  const bool valid = (uint32_t)lane < nv;
  const float *v = reinterpret_cast<const float *>(rec + a.vec_off);
  float4 *out = reinterpret_cast<float4 *>(a.blocks) + ((size_t)dst * a.dq) * kWave + lane;
  for (uint32_t qd = 0; qd < a.dq; ++qd) {
    float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
    const uint32_t e = qd * 4;
    if (valid) {
      if (e + 0 < a.dim) x.x = v[e + 0];
      if (e + 1 < a.dim) x.y = v[e + 1];
      if (e + 2 < a.dim) x.z = v[e + 2];
      if (e + 3 < a.dim) x.w = v[e + 3];
    }
    out[(size_t)qd * kWave] = x;
  }
  if (a.ext_ids) {
    uint64_t id = ~0ull;
    if (valid && a.id_off >= 0) id = *reinterpret_cast<const uint64_t *>(rec + a.id_off);
    a.ext_ids[(size_t)dst * kWave + lane] = id;
  }
}

// gather result vectors (include_vectors, api.rs:213-217) back to row-major
__global__ void gather_vectors_kernel(const float *blocks, uint32_t dq, uint32_t dim, const uint64_t *slots,
                                      uint64_t nres, float *V) {
  const uint64_t r = blockIdx.x;
  if (r >= nres) return;
  const uint64_t s = slots[r];
  for (uint32_t e = threadIdx.x; e < dim; e += blockDim.x) {
    float val = 0.0f;
    if (s != ~0ull) {
      const uint64_t blk = s / kWave, ln = s % kWave;
      val = blocks[((blk * dq + e / 4) * kWave + ln) * 4 + (e & 3)];
    }
    V[r * dim + e] = val;
  }
}

// rows of a row-major matrix -> lane-interleaved blocks (one lane per destination slot)
__global__ void __launch_bounds__(kBlockThreads) repack_rows_kernel(const float *src, uint32_t dim, uint32_t dq,
                                                                   const uint32_t *row_of_slot, uint64_t nslots,
                                                                   const uint64_t *id_of_row, float *blocks,
                                                                   uint64_t *ids_out) {
  const uint64_t s = (uint64_t)blockIdx.x * kBlockThreads + threadIdx.x;
  if (s >= nslots) return;
  const uint32_t row = row_of_slot[s];
  const bool valid = row != kNoPos;
  const float *v = src + (size_t)row * dim;
  float4 *out = reinterpret_cast<float4 *>(blocks) + ((s / kWave) * dq) * kWave + (s % kWave);
  for (uint32_t qd = 0; qd < dq; ++qd) {
    float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
    const uint32_t e = qd * 4;
    if (valid) {
      if (e + 0 < dim) x.x = v[e + 0];
      if (e + 1 < dim) x.y = v[e + 1];
      if (e + 2 < dim) x.z = v[e + 2];
      if (e + 3 < dim) x.w = v[e + 3];
    }
    out[(size_t)qd * kWave] = x;
  }
  if (ids_out) ids_out[s] = valid ? (id_of_row ? id_of_row[row] : (uint64_t)row) : ~0ull;
}

// plain pairwise distances in either order (vi_l2sq_pairs)
__global__ void l2sq_pairs_kernel(const float *a, const float *b, uint64_t n, uint32_t d, int order, float *out) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float *p = a + i * d, *c = b + i * d;
  out[i] = order == VI_ORDER_SCALAR ? l2sq_scalar_dev(p, c, d) : l2sq_lanes_dev(p, c, d);
}

// ------------------------------------------------------------------------------------------
// launch helpers
// ------------------------------------------------------------------------------------------
template <int QG, int ORDER, bool COARSE, bool DUMP>
vi_status launch_scan_t(const ScanArgs &a, uint32_t nitems_upper, hipStream_t st) {
  if (nitems_upper == 0) return VI_OK;
  const uint32_t grid = (nitems_upper + kWavesPerBlock - 1) / kWavesPerBlock;
  const size_t smem = (size_t)kWavesPerBlock * QG * a.dq * sizeof(float4);
  if (smem > 64 * 1024)
    VI_HIP(hipFuncSetAttribute((const void *)scan_kernel<QG, ORDER, COARSE, DUMP>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  hipLaunchKernelGGL((scan_kernel<QG, ORDER, COARSE, DUMP>), dim3(grid), dim3(kBlockThreads), smem, st, a);
  VI_HIP(hipGetLastError());
  return VI_OK;
}

}  // namespace

// Chosen so that QG*dq float4 per wave stays within the 160 KiB LDS of a CU at 4 waves/block.
int pick_qg(uint32_t dq, double avg_queries_per_unit, int order) {
  const int cap = order == VI_ORDER_LANES ? 4 : 8;
  // QG=8 halves the block re-reads but needs 154 VGPRs (3 waves/SIMD); QG=4 (102 VGPRs, 4 waves/SIMD)
  // measured 10 % faster on the C2 workload (profiles/r01_experiments.md)
  int qg = 1;
  if (avg_queries_per_unit >= 3.0) qg = 4;
  (void)cap;
  qg = std::min(qg, cap);
  while (qg > 1 && (size_t)kWavesPerBlock * qg * dq * sizeof(float4) > 96 * 1024) qg = (qg == 8) ? 4 : 1;
  return qg;
}

vi_status launch_scan(const ScanArgs &a, int qg, int order, bool coarse, uint32_t nitems_upper, hipStream_t st) {
#define VI_SCAN_CASE(QGV)                                                                                  \
  if (qg == QGV) {                                                                                         \
    constexpr int QL = QGV > 4 ? 4 : QGV; /* LANES order keeps 8 accumulators per pair: cap the group */   \
    if (a.dump_keys) {                                                                                     \
      if (order == VI_ORDER_SCALAR)                                                                        \
        return coarse ? launch_scan_t<QGV, VI_ORDER_SCALAR, true, true>(a, nitems_upper, st)               \
                      : launch_scan_t<QGV, VI_ORDER_SCALAR, false, true>(a, nitems_upper, st);             \
      return coarse ? launch_scan_t<QL, VI_ORDER_LANES, true, true>(a, nitems_upper, st)                   \
                    : launch_scan_t<QL, VI_ORDER_LANES, false, true>(a, nitems_upper, st);                 \
    }                                                                                                      \
    if (order == VI_ORDER_SCALAR)                                                                          \
      return coarse ? launch_scan_t<QGV, VI_ORDER_SCALAR, true, false>(a, nitems_upper, st)                \
                    : launch_scan_t<QGV, VI_ORDER_SCALAR, false, false>(a, nitems_upper, st);              \
    return coarse ? launch_scan_t<QL, VI_ORDER_LANES, true, false>(a, nitems_upper, st)                    \
                  : launch_scan_t<QL, VI_ORDER_LANES, false, false>(a, nitems_upper, st);                  \
  }
  VI_SCAN_CASE(1)
  VI_SCAN_CASE(2)
  VI_SCAN_CASE(4)
  VI_SCAN_CASE(8)
#undef VI_SCAN_CASE
  return fail(VI_ERR_OTHER, "unsupported query group %d", qg);
}

uint32_t coarse_splits(uint64_t nq, int qg, uint32_t nblk, uint32_t *bps) {
  const uint32_t nqg = (uint32_t)((nq + qg - 1) / qg);
  uint32_t S = std::max<uint32_t>(1, std::min<uint32_t>({(uint32_t)kMaxSelect, nblk, (4096 + nqg - 1) / nqg}));
  *bps = (std::max<uint32_t>(nblk, 1) + S - 1) / S;
  return std::max<uint32_t>(1, (nblk + *bps - 1) / *bps);
}

vi_status launch_repack_rows(const float *src, uint32_t dim, uint32_t dq, const uint32_t *row_of_slot,
                             uint64_t nslots, const uint64_t *id_of_row, float *blocks, uint64_t *ids_out,
                             hipStream_t st) {
  if (nslots == 0) return VI_OK;
  hipLaunchKernelGGL(repack_rows_kernel, dim3((uint32_t)((nslots + kBlockThreads - 1) / kBlockThreads)),
                     dim3(kBlockThreads), 0, st, src, dim, dq, row_of_slot, nslots, id_of_row, blocks, ids_out);
  VI_HIP(hipGetLastError());
  return VI_OK;
}

DeviceIndex::~DeviceIndex() {
  if (stream) (void)hipStreamDestroy(stream);
}

DeviceIndex::SearchContext::~SearchContext() {
  if (ws.hstats_pinned) (void)hipHostFree(ws.hstats_pinned);
  for (auto &e : ev)
    if (e) (void)hipEventDestroy(e);
  if (stream) (void)hipStreamDestroy(stream);
}

// the context held by the calling thread: set for the duration of device_index_search (a search of ANOTHER index from
// inside a search — the hierarchical k-means assignment does that — saves and restores it)
static thread_local DeviceIndex::SearchContext *t_ctx = nullptr;

DeviceIndex::SearchContext &DeviceIndex::cur() const { return *t_ctx; }

namespace {
struct ContextLease {
  const DeviceIndex &ix;
  DeviceIndex::SearchContext *mine = nullptr, *prev = nullptr;
  explicit ContextLease(const DeviceIndex &i) : ix(i) {}
  vi_status acquire() {
    std::unique_lock<std::mutex> lock(ix.mu);
    for (;;) {
      if (!ix.free_contexts.empty()) { mine = ix.free_contexts.back(); ix.free_contexts.pop_back(); break; }
      if ((int)ix.contexts.size() < DeviceIndex::kSearchContexts) {
        auto c = std::make_unique<DeviceIndex::SearchContext>();
        if (hipStreamCreateWithFlags(&c->stream, hipStreamDefault) != hipSuccess)
          return fail(VI_ERR_DEVICE, "hipStreamCreate failed: %s", hipGetErrorString(hipGetLastError()));
        for (auto &e : c->ev)
          if (hipEventCreate(&e) != hipSuccess) return fail(VI_ERR_DEVICE, "hipEventCreate failed");
        mine = c.get();
        ix.contexts.push_back(std::move(c));
        break;
      }
      ix.cv.wait(lock);
    }
    prev = t_ctx;
    t_ctx = mine;
    return VI_OK;
  }
  ~ContextLease() {
    if (!mine) return;
    t_ctx = prev;
    {
      std::lock_guard<std::mutex> lock(ix.mu);
      ix.last_stats = mine->stats;
      ix.free_contexts.push_back(mine);
    }
    ix.cv.notify_one();
  }
};
}  // namespace

// ------------------------------------------------------------------------------------------
// upload
// ------------------------------------------------------------------------------------------
static vi_status repack_upload(const uint8_t *host_src, size_t src_bytes, const std::vector<uint64_t> &blk_src,
                               const std::vector<uint32_t> &blk_nv, const std::vector<uint32_t> &blk_dst,
                               uint32_t dim, uint32_t dq, uint32_t stride, uint32_t vec_off, int32_t id_off,
                               float *blocks, uint64_t *ext_ids, hipStream_t st) {
  const uint32_t nb = (uint32_t)blk_src.size();
  if (nb == 0) return VI_OK;
  DevBuf<uint8_t> img;
  DevBuf<uint64_t> dsrc;
  DevBuf<uint32_t> dnv, ddst;
  VI_TRY(img.reserve(src_bytes + 16));
  VI_TRY(dsrc.reserve(nb));
  VI_TRY(dnv.reserve(nb));
  VI_TRY(ddst.reserve(nb));
  VI_HIP(hipMemcpyAsync(img.p, host_src, src_bytes, hipMemcpyHostToDevice, st));
  VI_HIP(hipMemcpyAsync(dsrc.p, blk_src.data(), nb * sizeof(uint64_t), hipMemcpyHostToDevice, st));
  VI_HIP(hipMemcpyAsync(dnv.p, blk_nv.data(), nb * sizeof(uint32_t), hipMemcpyHostToDevice, st));
  VI_HIP(hipMemcpyAsync(ddst.p, blk_dst.data(), nb * sizeof(uint32_t), hipMemcpyHostToDevice, st));
  RepackArgs a{img.p, dsrc.p, dnv.p, ddst.p, nb, dim, dq, stride, vec_off, id_off, blocks, ext_ids};
  hipLaunchKernelGGL(repack_kernel, dim3((nb + kWavesPerBlock - 1) / kWavesPerBlock), dim3(kBlockThreads), 0, st, a);
  VI_HIP(hipGetLastError());
  VI_HIP(hipStreamSynchronize(st));  // img is freed on return
  return VI_OK;
}

static vi_status init_device_index(DeviceIndex *ix, int device, uint32_t dim, uint64_t nlists);

std::vector<uint32_t> shard_owners(const std::vector<uint64_t> &shard_bytes, uint32_t world) {
  const size_t ns = shard_bytes.size();
  std::vector<uint32_t> order(ns), owner(ns, 0);
  for (size_t i = 0; i < ns; ++i) order[i] = (uint32_t)i;
  std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return shard_bytes[a] > shard_bytes[b]; });
  std::vector<uint64_t> load(std::max<uint32_t>(world, 1), 0);
  for (uint32_t s : order) {
    uint32_t best = 0;
    for (uint32_t r = 1; r < load.size(); ++r)
      if (load[r] < load[best]) best = r;
    owner[s] = best;
    load[best] += shard_bytes[s];
  }
  return owner;
}

vi_status device_index_load(const IndexMeta &meta, const std::string &shards_dir, int device, int rank, int world,
                            int placement, DeviceIndex *ix) {
  VI_TRY(init_device_index(ix, device, meta.dimension, meta.k()));
  const uint32_t dim = ix->dim, dq = ix->dq;
  const uint64_t k = ix->nlists;

  // ---- centroid table ----
  {
    const uint64_t nb = (k + kWave - 1) / kWave;
    ix->centroids.dq = dq;
    ix->centroids.nblocks = nb;
    VI_TRY(ix->centroids.blocks.reserve(std::max<uint64_t>(1, nb) * dq * kWave * 4));
    std::vector<uint64_t> src(nb);
    std::vector<uint32_t> nv(nb), dst(nb);
    for (uint64_t b = 0; b < nb; ++b) {
      src[b] = b * kWave * (uint64_t)dim * 4;
      nv[b] = (uint32_t)std::min<uint64_t>(kWave, k - b * kWave);
      dst[b] = (uint32_t)b;
    }
    VI_TRY(repack_upload((const uint8_t *)meta.centroids.data(), meta.centroids.size() * 4, src, nv, dst, dim, dq,
                         dim * 4, 0, -1, ix->centroids.blocks.p, nullptr, ix->stream));
  }

  // ---- lists: every rank maps all shard files and keeps a STRIPE of every list resident: block b (64 vectors)
  //      of a list lives on rank b % world.  Placement of whole lists (or whole shard files, the reference's
  //      grouping) cannot balance: on the bench workload two lists carry 52 % of the scan work, and the
  //      reference's super-k-means puts neighbouring lists into one shard (92 % / 8 % of the work at world 2).
  //      With stripes every rank scans 1/world of every probed list whatever the skew.  A local position p maps
  //      back to the list position ((p / 64) * world + rank) * 64 + p % 64 (tie keys, stripe_tie_kernel). ----
  uint64_t nshards = 0;
  for (uint64_t c = 0; c < k; ++c) nshards = std::max(nshards, meta.c2s[c] + 1);
  ix->nshards = nshards;
  std::vector<uint32_t> h_first(k, 0), h_len(k, 0), h_shard(k, 0);
  for (uint64_t c = 0; c < k; ++c) h_shard[c] = (uint32_t)meta.c2s[c];
  std::vector<std::unique_ptr<ShardFile>> files(nshards);
  const bool part = world > 1;
  uint64_t total_blocks = 0, total_vec = 0;
  for (uint64_t s = 0; s < nshards; ++s) {
    auto f = std::make_unique<ShardFile>();
    // a missing/corrupt shard is skipped, as search does (`if let Ok`, ivf_index.rs:253-254)
    if (f->open(shards_dir, s) != VI_OK || f->dim() != dim) continue;
    files[s] = std::move(f);
  }
  for (uint64_t c = 0; c < k; ++c) {
    const uint64_t s = meta.c2s[c];
    if (s >= nshards || !files[s]) continue;
    const ShardListView *lv = files[s]->find(c);
    if (!lv) { files[s].reset(); continue; }  // NotFound fails the whole shard read (shards.rs:257-265)
  }
  // placement 1 (north_star's rule): whole shard files to ranks, greedy by bytes; this rank then holds every block of
  // the lists of its shards (W = 1 below) and nothing of the others
  std::vector<uint32_t> owner;
  const bool by_shard = part && placement == 1;
  if (by_shard) {
    std::vector<uint64_t> bytes(nshards, 0);
    for (uint64_t c = 0; c < k; ++c) {
      const uint64_t s = meta.c2s[c];
      if (s >= nshards || !files[s]) continue;
      if (const ShardListView *lv = files[s]->find(c)) bytes[s] += (uint64_t)lv->num_vectors * record_stride(dim);
    }
    owner = shard_owners(bytes, (uint32_t)world);
    for (uint64_t s = 0; s < nshards; ++s)
      if (owner[s] != (uint32_t)rank) files[s].reset();  // (as if the file were not there: its lists stay empty here)
  }
  const uint32_t W = (part && !by_shard) ? (uint32_t)world : 1u, R = (part && !by_shard) ? (uint32_t)rank : 0u;
  ix->stripe_world = W;
  ix->stripe_rank = R;
  // this rank's blocks of a list of n vectors: b = R, R+W, ... ; the last block of the list may be partial
  auto local_blocks = [&](uint32_t n) { const uint32_t nb = (n + kWave - 1) / kWave; return nb > R ? (nb - R + W - 1) / W : 0u; };
  auto local_len = [&](uint32_t n) {
    const uint32_t nb = (n + kWave - 1) / kWave, lb = local_blocks(n);
    if (lb == 0) return 0u;
    const uint32_t last = R + (lb - 1) * W;  // this rank's last block
    return (lb - 1) * kWave + (last == nb - 1 ? n - last * kWave : (uint32_t)kWave);
  };
  for (uint64_t c = 0; c < k; ++c) {
    const uint64_t s = meta.c2s[c];
    if (s >= nshards || !files[s]) continue;
    const ShardListView *lv = files[s]->find(c);
    h_first[c] = (uint32_t)total_blocks;
    h_len[c] = local_len(lv->num_vectors);
    total_blocks += local_blocks(lv->num_vectors);
    total_vec += h_len[c];
  }
  if (total_blocks >= 0xFFFFFFFFull / kWave) return fail(VI_ERR_OTHER, "index too large for 32-bit slot ids");
  ix->nvec_resident = total_vec;
  ix->lists.dq = dq;
  ix->lists.nblocks = total_blocks;
  VI_TRY(ix->lists.blocks.reserve(std::max<uint64_t>(1, total_blocks) * dq * kWave * 4));
  VI_TRY(ix->ext_ids.reserve(std::max<uint64_t>(1, total_blocks) * kWave));
  const uint32_t stride = (uint32_t)record_stride(dim);
  for (uint64_t s = 0; s < nshards; ++s) {
    if (!files[s]) continue;
    const ShardFile &f = *files[s];
    std::vector<uint64_t> src;
    std::vector<uint32_t> nv, dst;
    const uint8_t *lo = nullptr, *hi = nullptr;
    for (uint32_t i = 0; i < f.num_lists(); ++i) {
      const ShardListView &lv = f.list(i);
      // (a file with duplicate centroid ids: only the first entry exists for the reference's linear find,
      // shards.rs:257-265 — a later one must not be uploaded over the blocks sized from the first)
      if (lv.centroid_id >= k || meta.c2s[lv.centroid_id] != s || lv.num_vectors == 0 || f.find(lv.centroid_id) != &lv) continue;
      if (!lo || lv.records < lo) lo = lv.records;
      const uint8_t *end = lv.records + (uint64_t)lv.num_vectors * stride;
      if (!hi || end > hi) hi = end;
    }
    if (!lo) continue;
    for (uint32_t i = 0; i < f.num_lists(); ++i) {
      const ShardListView &lv = f.list(i);
      if (lv.centroid_id >= k || meta.c2s[lv.centroid_id] != s || lv.num_vectors == 0 || f.find(lv.centroid_id) != &lv) continue;
      const uint32_t nb = (lv.num_vectors + kWave - 1) / kWave;
      for (uint32_t b = R; b < nb; b += W) {
        src.push_back((uint64_t)(lv.records - lo) + (uint64_t)b * kWave * stride);
        nv.push_back(std::min<uint32_t>(kWave, lv.num_vectors - b * kWave));
        dst.push_back(h_first[lv.centroid_id] + (b - R) / W);
        if (dst.back() >= total_blocks) return fail(VI_ERR_INVALID_DATA, "shard_%llu.bin: list blocks exceed the index layout", (unsigned long long)s);
      }
    }
    VI_TRY(repack_upload(lo, (size_t)(hi - lo), src, nv, dst, dim, dq, stride, (uint32_t)kVectorMetaBytes, 8,
                         ix->lists.blocks.p, ix->ext_ids.p, ix->stream));
  }
  VI_TRY(ix->list_first_block.reserve(std::max<uint64_t>(1, k)));
  VI_TRY(ix->list_len.reserve(std::max<uint64_t>(1, k)));
  VI_TRY(ix->list_shard.reserve(std::max<uint64_t>(1, k)));
  if (k) {
    VI_HIP(hipMemcpy(ix->list_first_block.p, h_first.data(), k * 4, hipMemcpyHostToDevice));
    VI_HIP(hipMemcpy(ix->list_len.p, h_len.data(), k * 4, hipMemcpyHostToDevice));
    VI_HIP(hipMemcpy(ix->list_shard.p, h_shard.data(), k * 4, hipMemcpyHostToDevice));
  }
  return compute_slot_norms(ix);
}

static vi_status init_device_index(DeviceIndex *ix, int device, uint32_t dim, uint64_t nlists) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(VI_ERR_DEVICE, "no HIP device visible: libvi_amd never falls back to the CPU");
  if (device < 0 || device >= ndev) return fail(VI_ERR_DEVICE, "device %d out of range (%d visible)", device, ndev);
  VI_HIP(hipSetDevice(device));
  ix->device = device;
  ix->dim = dim;
  ix->dq = layout_dq(dim);
  ix->nlists = nlists;
  if (!ix->stream) VI_HIP(hipStreamCreateWithFlags(&ix->stream, hipStreamDefault));
  return VI_OK;
}

vi_status init_device_index_pub(DeviceIndex *ix, int device, uint32_t dim, uint64_t nlists) {
  return init_device_index(ix, device, dim, nlists);
}

vi_status device_index_from_rows(int device, int order, uint32_t dim, const float *table_dev, uint64_t ntable,
                                 const float *rows_dev, const std::vector<uint64_t> &list_off,
                                 const std::vector<uint32_t> &member_rows, const uint64_t *ids_dev,
                                 const std::vector<uint32_t> *list_shard, DeviceIndex *ix) {
  const uint64_t nlists = list_off.size() - 1;
  if (nlists != ntable) return fail(VI_ERR_INVALID_INPUT, "table rows must equal list count");
  VI_TRY(init_device_index(ix, device, dim, nlists));
  ix->order = order;
  const uint32_t dq = ix->dq;
  hipStream_t st = ix->stream;
  // coarse table
  {
    const uint64_t nb = (ntable + kWave - 1) / kWave;
    ix->centroids.dq = dq;
    ix->centroids.nblocks = nb;
    VI_TRY(ix->centroids.blocks.reserve(std::max<uint64_t>(1, nb) * dq * kWave * 4));
    std::vector<uint32_t> ros(nb * kWave, kNoPos);
    for (uint64_t i = 0; i < ntable; ++i) ros[i] = (uint32_t)i;
    DevBuf<uint32_t> dros;
    VI_TRY(dros.reserve(ros.size()));
    VI_HIP(hipMemcpyAsync(dros.p, ros.data(), ros.size() * 4, hipMemcpyHostToDevice, st));
    VI_TRY(launch_repack_rows(table_dev, dim, dq, dros.p, ros.size(), nullptr, ix->centroids.blocks.p, nullptr, st));
    VI_HIP(hipStreamSynchronize(st));
  }
  std::vector<uint32_t> h_first(nlists, 0), h_len(nlists, 0), h_shard(nlists, 0);
  uint64_t total_blocks = 0;
  for (uint64_t l = 0; l < nlists; ++l) {
    const uint64_t len = list_off[l + 1] - list_off[l];
    h_first[l] = (uint32_t)total_blocks;
    h_len[l] = (uint32_t)len;
    h_shard[l] = list_shard ? (*list_shard)[l] : (uint32_t)l;
    total_blocks += (len + kWave - 1) / kWave;
  }
  if (total_blocks >= 0xFFFFFFFFull / kWave) return fail(VI_ERR_OTHER, "index too large for 32-bit slot ids");
  ix->nvec_resident = member_rows.size();
  ix->nshards = 0;
  for (uint64_t l = 0; l < nlists; ++l) ix->nshards = std::max<uint64_t>(ix->nshards, (uint64_t)h_shard[l] + 1);
  ix->lists.dq = dq;
  ix->lists.nblocks = total_blocks;
  VI_TRY(ix->lists.blocks.reserve(std::max<uint64_t>(1, total_blocks) * dq * kWave * 4));
  VI_TRY(ix->ext_ids.reserve(std::max<uint64_t>(1, total_blocks) * kWave));
  {
    std::vector<uint32_t> ros(total_blocks * kWave, kNoPos);
    for (uint64_t l = 0; l < nlists; ++l)
      for (uint64_t e = list_off[l]; e < list_off[l + 1]; ++e)
        ros[(uint64_t)h_first[l] * kWave + (e - list_off[l])] = member_rows[e];
    DevBuf<uint32_t> dros;
    VI_TRY(dros.reserve(ros.size()));
    if (!ros.empty()) VI_HIP(hipMemcpyAsync(dros.p, ros.data(), ros.size() * 4, hipMemcpyHostToDevice, st));
    VI_TRY(launch_repack_rows(rows_dev, dim, dq, dros.p, ros.size(), ids_dev, ix->lists.blocks.p, ix->ext_ids.p, st));
    VI_HIP(hipStreamSynchronize(st));
  }
  VI_TRY(ix->list_first_block.reserve(std::max<uint64_t>(1, nlists)));
  VI_TRY(ix->list_len.reserve(std::max<uint64_t>(1, nlists)));
  VI_TRY(ix->list_shard.reserve(std::max<uint64_t>(1, nlists)));
  if (nlists) {
    VI_HIP(hipMemcpy(ix->list_first_block.p, h_first.data(), nlists * 4, hipMemcpyHostToDevice));
    VI_HIP(hipMemcpy(ix->list_len.p, h_len.data(), nlists * 4, hipMemcpyHostToDevice));
    VI_HIP(hipMemcpy(ix->list_shard.p, h_shard.data(), nlists * 4, hipMemcpyHostToDevice));
  }
  return compute_slot_norms(ix);
}

vi_status search_filter_pipeline(const DeviceIndex &ix, const float *Qd, uint64_t nq, uint64_t k, uint32_t P, uint32_t K,
                                 float *Dd, int64_t *Id, uint64_t *Td, uint64_t *slots, uint32_t *counts, hipStream_t st,
                                 int timing_level, const uint32_t *probes_in, const uint32_t *order_in);
vi_status coarse_only_filter(const DeviceIndex &ix, const float *Qd, uint64_t nq, uint32_t P, hipStream_t st);
bool filter_path_applicable(const DeviceIndex &ix, uint64_t nq, uint64_t k, uint32_t P);

vi_status device_index_search_generic(const DeviceIndex &ix, const float *Qd, uint64_t nq, uint64_t k, uint32_t P,
                                      float *Dd, int64_t *Id, uint64_t *Td, uint64_t *slots, uint32_t *counts,
                                      hipStream_t st, const uint32_t *probes_in, const uint32_t *order_in);
vi_status generic_probe_export(const DeviceIndex &ix, const float *Qd, uint64_t nq, uint32_t P, hipStream_t st);

// ------------------------------------------------------------------------------------------
// pipeline stages
// ------------------------------------------------------------------------------------------
// histogram (ws.cnt) -> totals -> offsets of the lists in pairs / items / records -> scatter cursors
static vi_status launch_group_scan(const DeviceIndex &ix, uint32_t qg, uint32_t segb0, uint32_t *tile_start, hipStream_t st,
                                   bool reset_stats = false, const uint32_t *qtot = nullptr, uint32_t nq = 0, uint32_t *qoff = nullptr) {
  SearchWorkspace &ws = ix.cur().ws;
  const uint32_t nlists = (uint32_t)ix.nlists;
  VI_TRY(ws.list_tot.reserve(std::max<uint32_t>(1, nlists)));
  const dim3 grid((nlists + 255) / 256), block(256);
  hipLaunchKernelGGL(list_totals_kernel, grid, block, 0, st, ws.cnt.p, nlists, ws.list_tot.p, reset_stats ? ws.stats.p : nullptr);
  if (qtot)  // (+ the queries' record offsets: a second workgroup of the same launch)
    hipLaunchKernelGGL(group_prepare_kernel, dim3(2), dim3(1024), 0, st, ws.list_tot.p, ix.list_len.p, nlists, qg, segb0,
                       ws.seg_start.p, ws.item_start.p, ws.segrun_start.p, ws.stats.p, tile_start, qtot, nq, qoff);
  else
    hipLaunchKernelGGL(group_scan_kernel, dim3(1), dim3(1024), 0, st, ws.list_tot.p, ix.list_len.p, nlists, qg, segb0,
                       ws.seg_start.p, ws.item_start.p, ws.segrun_start.p, ws.stats.p, tile_start);
  hipLaunchKernelGGL(cursor_kernel, grid, block, 0, st, ws.cnt.p, ws.seg_start.p, nlists, ws.cnt.p + subbin_words(nlists));
  VI_HIP(hipGetLastError());
  return VI_OK;
}

// 1+2: coarse scan over the centroid table, merge -> ws.probes / ws.gorder, histogram in ws.cnt
vi_status stage_coarse(const DeviceIndex &ix, const float *Qd, uint64_t nq, uint32_t P, hipStream_t st) {
  SearchWorkspace &ws = ix.cur().ws;
  const uint32_t dim = ix.dim, dq = ix.dq;
  const uint64_t nlists = ix.nlists;
  VI_TRY(ws.cnt.reserve(2 * subbin_words(nlists)));
  VI_HIP(hipMemsetAsync(ws.cnt.p, 0, subbin_words(nlists) * sizeof(uint32_t), st));
  const uint32_t nblk_c = (uint32_t)ix.centroids.nblocks;
  const int qg_c = pick_qg(dq, (double)nq, ix.order);
  const uint32_t nqg = (uint32_t)((nq + qg_c - 1) / qg_c);
  uint32_t bps = 0;
  const uint32_t S = coarse_splits(nq, qg_c, nblk_c, &bps);
  VI_TRY(ws.crun_dist.reserve(nq * S * P));
  VI_TRY(ws.crun_pos.reserve(nq * S * P));
  {
    ScanArgs a{};
    a.blocks = (const float4 *)ix.centroids.blocks.p; a.dq = dq; a.dim = dim; a.Q = Qd; a.nq = (uint32_t)nq;
    a.K = P; a.run_dist = ws.crun_dist.p; a.run_pos = ws.crun_pos.p;
    a.nvec = (uint32_t)nlists; a.S = S; a.bps = bps;
    VI_TRY(launch_scan(a, qg_c, ix.order, true, nqg * S, st));
  }
  VI_TRY(ws.probes.reserve(nq * P));
  VI_TRY(ws.gorder.reserve(nq * P));
  CoarseMergeArgs a{ws.crun_dist.p, ws.crun_pos.p, (uint32_t)nq, S, P, ix.list_shard.p, ix.list_len.p,
                    ws.probes.p, ws.gorder.p, ws.cnt.p, (uint32_t)nlists};
  hipLaunchKernelGGL(coarse_merge_kernel, dim3((uint32_t)((nq + kWavesPerBlock - 1) / kWavesPerBlock)),
                     dim3(kBlockThreads), 0, st, a);
  VI_HIP(hipGetLastError());
  return VI_OK;
}

// exact-order VALU pipeline: coarse -> group -> list scan -> final merge
// probes given by the caller (multi-GPU: coarse step done elsewhere): probe lists + candidate-order ranks into
// the workspace, and the per-list histogram the grouping scan starts from
vi_status adopt_probes(const DeviceIndex &ix, uint64_t nq, uint32_t P, const uint32_t *probes_in, const uint32_t *order_in,
                       bool histogram, hipStream_t st) {
  SearchWorkspace &ws = ix.cur().ws;
  const uint64_t nlists = ix.nlists;
  VI_TRY(ws.probes.reserve(nq * P));
  VI_TRY(ws.gorder.reserve(nq * P));
  VI_TRY(ws.probe_flag.reserve(1));
  const uint32_t total = (uint32_t)(nq * P);
  VI_HIP(hipMemsetAsync(ws.probe_flag.p, 0, 4, st));
  hipLaunchKernelGGL(validate_probes_kernel, dim3((total + 255) / 256), dim3(256), 0, st, probes_in, order_in, total, P,
                     (uint32_t)nlists, ws.probe_flag.p);
  VI_HIP(hipGetLastError());
  uint32_t bad = 0;
  VI_HIP(hipMemcpyAsync(&bad, ws.probe_flag.p, 4, hipMemcpyDeviceToHost, st));
  VI_HIP(hipMemcpyAsync(ws.probes.p, probes_in, nq * P * 4, hipMemcpyDeviceToDevice, st));
  VI_HIP(hipMemcpyAsync(ws.gorder.p, order_in, nq * P * 4, hipMemcpyDeviceToDevice, st));
  VI_HIP(hipStreamSynchronize(st));
  if (bad)
    return fail(VI_ERR_INVALID_INPUT, "probe lists out of range: every probe must be < %llu (or the empty marker 0xFFFFFFFF "
                "after the last real probe of a row) and every order < n_probe_eff", (unsigned long long)nlists);
  if (histogram) {
    VI_TRY(ws.cnt.reserve(2 * subbin_words(nlists)));
    VI_HIP(hipMemsetAsync(ws.cnt.p, 0, subbin_words(nlists) * sizeof(uint32_t), st));
    hipLaunchKernelGGL(histogram_kernel, dim3((total + 255) / 256), dim3(256), 0, st, ws.probes.p, ix.list_len.p,
                       (uint32_t)nlists, total, P, ws.cnt.p);
    VI_HIP(hipGetLastError());
  }
  return VI_OK;
}

vi_status search_valu_pipeline(const DeviceIndex &ix, const float *Qd, uint64_t nq, uint64_t k, uint32_t P, uint32_t K,
                               float *Dd, int64_t *Id, uint64_t *Td, uint64_t *slots, uint32_t *counts, hipStream_t st,
                               int timing_level, const uint32_t *probes_in, const uint32_t *order_in) {
  const bool timing = timing_level == 1, rank_timing = timing_level != 0;
  SearchWorkspace &ws = ix.cur().ws;
  vi_search_stats &stt = ix.cur().stats;
  const uint32_t dim = ix.dim, dq = ix.dq;
  const uint64_t nlists = ix.nlists;
  VI_TRY(ws.run_dist.reserve(nq * P * K));
  VI_TRY(ws.run_pos.reserve(nq * P * K));
  // runs of lists that are not resident here (other rank / unreadable shard) stay empty
  VI_HIP(hipMemsetAsync(ws.run_pos.p, 0xFF, nq * P * K * sizeof(uint32_t), st));

  if (timing) VI_HIP(hipEventRecord(ix.cur().ev[0], st));
  if (probes_in) VI_TRY(adopt_probes(ix, nq, P, probes_in, order_in, true, st));
  else VI_TRY(stage_coarse(ix, Qd, nq, P, st));
  if (timing) VI_HIP(hipEventRecord(ix.cur().ev[1], st));
  // ---- 3. group (query,probe) pairs by list ----
  const double avg_q_per_list = (double)nq * P / (double)std::max<uint64_t>(1, nlists);
  int qg_l = pick_qg(dq, avg_q_per_list, ix.order);
  {
    const uint32_t f = env_u32("VI_FORCE_QG", 0);
    if (f == 1 || f == 2 || f == 4 || (f == 8 && ix.order == VI_ORDER_SCALAR)) qg_l = (int)f;
  }
  const uint32_t kSegBlocks = std::max<uint32_t>(1, env_u32("VI_SEG_BLOCKS", kSegBlocksDefault));
  VI_TRY(ws.seg_start.reserve(nlists + 1));
  VI_TRY(ws.item_start.reserve(nlists + 1));
  VI_TRY(ws.pairs.reserve(nq * P));
  VI_TRY(ws.segrun_start.reserve(nlists + 1));
  VI_HIP(hipMemsetAsync(ws.stats.p, 0, 6 * sizeof(uint64_t), st));  // [6] .. [11] belong to the MFMA path's select
  VI_TRY(launch_group_scan(ix, (uint32_t)qg_l, kSegBlocks, nullptr, st));
  // exact work-item / segment-run counts size the scan grid and its scratch; the host waits for them while the
  // scatter runs
  uint64_t hstats[3] = {0, 0, 0};
  VI_HIP(hipMemcpyAsync(hstats, ws.stats.p, sizeof(hstats), hipMemcpyDeviceToHost, st));
  VI_HIP(hipEventRecord(ix.cur().ev[5], st));
  {
    const uint32_t total = (uint32_t)(nq * P);
    hipLaunchKernelGGL(group_scatter_kernel, dim3((total + 255) / 256), dim3(256), 0, st, ws.probes.p,
                       ix.list_len.p, (uint32_t)nlists, P, ws.cnt.p + subbin_words(nlists), ws.pairs.p, total,
                       ws.seg_start.p, (uint32_t *)nullptr);
    VI_HIP(hipGetLastError());
  }
  VI_HIP(hipEventSynchronize(ix.cur().ev[5]));
  stt.scanned_vectors = hstats[0];
  stt.scan_items = hstats[1];
  const uint64_t nsegruns = hstats[2];
  VI_TRY(ws.seg_run_dist.reserve(nsegruns * K));
  VI_TRY(ws.seg_run_pos.reserve(nsegruns * K));
  if (rank_timing) VI_HIP(hipEventRecord(ix.cur().ev[2], st));
  // ---- 4. list scan ----
  {
    ScanArgs a{};
    a.blocks = (const float4 *)ix.lists.blocks.p; a.dq = dq; a.dim = dim; a.Q = Qd; a.nq = (uint32_t)nq;
    a.K = K; a.run_dist = ws.run_dist.p; a.run_pos = ws.run_pos.p;
    a.first_block = ix.list_first_block.p; a.list_len = ix.list_len.p; a.item_start = ws.item_start.p;
    a.seg_start = ws.seg_start.p; a.pairs = ws.pairs.p; a.nlists = (uint32_t)nlists; a.P = P;
    a.segb0 = kSegBlocks; a.segrun_start = ws.segrun_start.p;
    a.seg_run_dist = ws.seg_run_dist.p; a.seg_run_pos = ws.seg_run_pos.p;
    VI_TRY(launch_scan(a, qg_l, ix.order, false, (uint32_t)hstats[1], st));
    if (nsegruns) {
      SegMergeArgs m{ws.seg_start.p, ws.segrun_start.p, ix.list_len.p, ws.pairs.p, (uint32_t)nlists, kSegBlocks, K,
                     ws.seg_run_dist.p, ws.seg_run_pos.p, ws.run_dist.p, ws.run_pos.p};
      const uint32_t npairs = (uint32_t)(nq * P);
      hipLaunchKernelGGL(seg_merge_kernel, dim3((npairs + kWavesPerBlock - 1) / kWavesPerBlock), dim3(kBlockThreads),
                         0, st, m);
      VI_HIP(hipGetLastError());
    }
  }
  if (rank_timing) VI_HIP(hipEventRecord(ix.cur().ev[3], st));
  // ---- 5. final merge ----
  {
    FinalMergeArgs a{ws.run_dist.p, ws.run_pos.p, (uint32_t)nq, P, K, (uint32_t)k, ws.probes.p, ws.gorder.p,
                     ix.list_first_block.p, ix.ext_ids.p, Dd, Id, Td, slots, counts};
    hipLaunchKernelGGL(final_merge_kernel, dim3((uint32_t)((nq + kWavesPerBlock - 1) / kWavesPerBlock)),
                       dim3(kBlockThreads), 0, st, a);
    VI_HIP(hipGetLastError());
  }
  if (timing) VI_HIP(hipEventRecord(ix.cur().ev[4], st));
  return VI_OK;
}

// ------------------------------------------------------------------------------------------
// search pipeline (fast path: n_probe_eff <= 64 and k <= 64)
// ------------------------------------------------------------------------------------------
// striped lists: local position -> position in the whole list, so that the merge over ranks sees the
// reference's candidate order
__global__ void stripe_tie_kernel(uint64_t *tie, uint64_t n, uint32_t rank, uint32_t world) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t t = tie[i];
  if (t == ~0ull) return;
  const uint32_t p = (uint32_t)t;
  tie[i] = (t & 0xFFFFFFFF00000000ull) | (uint64_t)(((p >> 6) * world + rank) * 64u + (p & 63u));
}

vi_status device_index_search(const DeviceIndex &ix, const SearchIO &io) {
  VI_HIP(hipSetDevice(ix.device));
  ContextLease lease(ix);
  VI_TRY(lease.acquire());
  hipStream_t st = ix.cur().stream;
  SearchWorkspace &ws = ix.cur().ws;
  const uint64_t nq = io.nq, k = io.k;
  const uint32_t dim = ix.dim, dq = ix.dq;
  const uint64_t nlists = ix.nlists;
  if (nq == 0) return VI_OK;
  if (nq * std::max<uint64_t>(k, 64) > 0x7FFFFFFFull) return fail(VI_ERR_INVALID_INPUT, "batch too large: split nq");
  const uint32_t P = (uint32_t)std::min<uint64_t>(io.n_probe, nlists);  // take(n_probe) (ivf_index.rs:216-220)
  const uint32_t K = (uint32_t)std::min<uint64_t>(k, kMaxSelect);
  // k in (64, 128]: the MFMA engine's select keeps two entries per lane; the VALU engine's wave top-k stops at 64
  const bool generic = P > kMaxSelect || k > 2 * kMaxSelect || (k > kMaxSelect && !filter_path_applicable(ix, nq, k, P)) ||
                       env_u32("VI_FORCE_GENERIC", 0) != 0;

  // ---- outputs / queries on device ----
  const float *Qd = io.queries;
  float *Dd = io.D;
  int64_t *Id = io.I;
  uint64_t *Td = io.tie;
  if (!io.on_device) {
    VI_TRY(ws.q.reserve(nq * dim));
    VI_TRY(ws.D.reserve(nq * k));
    VI_TRY(ws.I.reserve(nq * k));
    VI_HIP(hipMemcpyAsync(ws.q.p, io.queries, nq * dim * sizeof(float), hipMemcpyHostToDevice, st));
    Qd = ws.q.p; Dd = ws.D.p; Id = ws.I.p; Td = nullptr;
  }
  VI_TRY(ws.counts.reserve(nq));
  VI_TRY(ws.stats.reserve(160));
  uint64_t *slots = nullptr;
  if (io.V) { VI_TRY(ws.slots.reserve(nq * k)); slots = ws.slots.p; }

  vi_search_stats &stt = ix.cur().stats;
  stt = vi_search_stats{};
  stt.nq = nq; stt.k = k; stt.n_probe_eff = P; stt.coarse_candidates = nq * nlists;

  if (P == 0 || nlists == 0) {  // empty index: every query has zero results
    std::vector<float> hD(nq * k, INFINITY);
    std::vector<int64_t> hI(nq * k, -1);
    if (io.on_device) {
      VI_HIP(hipMemcpy(io.D, hD.data(), hD.size() * 4, hipMemcpyHostToDevice));
      VI_HIP(hipMemcpy(io.I, hI.data(), hI.size() * 8, hipMemcpyHostToDevice));
    } else {
      std::memcpy(io.D, hD.data(), hD.size() * 4);
      std::memcpy(io.I, hI.data(), hI.size() * 8);
      if (io.counts) std::memset(io.counts, 0, nq * 8);
      if (io.V) std::memset(io.V, 0, nq * k * dim * 4);
    }
    return VI_OK;
  }

  const bool use_filter = !generic && filter_path_applicable(ix, nq, k, P);
  const int timing = generic ? 0 : ix.timing;
  if (io.probes_out) {  // coarse step only (multi-GPU: this rank's slice of the queries)
    if (P > kMaxSelect) VI_TRY(generic_probe_export(ix, Qd, nq, P, st));  // any n_probe: every coarse distance, sorted
    else if (filter_path_applicable(ix, nq, 1, P) && nq >= 256 && nlists >= 1024) VI_TRY(coarse_only_filter(ix, Qd, nq, P, st));
    else VI_TRY(stage_coarse(ix, Qd, nq, P, st));
    VI_HIP(hipMemcpyAsync(io.probes_out, ws.probes.p, nq * P * 4, hipMemcpyDeviceToDevice, st));
    VI_HIP(hipMemcpyAsync(io.order_out, ws.gorder.p, nq * P * 4, hipMemcpyDeviceToDevice, st));
    VI_HIP(hipStreamSynchronize(st));
    return VI_OK;
  }
  if (generic) {  // (k > 128 or n_probe > 64, with the caller's probe lists too)
    VI_TRY(device_index_search_generic(ix, Qd, nq, k, P, Dd, Id, Td, slots, ws.counts.p, st, io.probes_in, io.order_in));
  } else if (use_filter) {
    VI_TRY(search_filter_pipeline(ix, Qd, nq, k, P, K, Dd, Id, Td, slots, ws.counts.p, st, timing, io.probes_in, io.order_in));
  } else {
    VI_TRY(search_valu_pipeline(ix, Qd, nq, k, P, K, Dd, Id, Td, slots, ws.counts.p, st, timing, io.probes_in, io.order_in));
  }

  if (ix.stripe_world > 1 && Td) {
    hipLaunchKernelGGL(stripe_tie_kernel, dim3((uint32_t)((nq * k + 255) / 256)), dim3(256), 0, st, Td, nq * k,
                       ix.stripe_rank, ix.stripe_world);
    VI_HIP(hipGetLastError());
  }
  // ---- results ----
  if (!io.on_device) {
    VI_HIP(hipMemcpyAsync(io.D, Dd, nq * k * sizeof(float), hipMemcpyDeviceToHost, st));
    VI_HIP(hipMemcpyAsync(io.I, Id, nq * k * sizeof(int64_t), hipMemcpyDeviceToHost, st));
    if (io.V) {
      VI_TRY(ws.V.reserve(nq * k * dim));
      hipLaunchKernelGGL(gather_vectors_kernel, dim3((uint32_t)(nq * k)), dim3(64), 0, st, ix.lists.blocks.p, dq, dim,
                         slots, nq * k, ws.V.p);
      VI_HIP(hipGetLastError());
      VI_HIP(hipMemcpyAsync(io.V, ws.V.p, nq * k * dim * sizeof(float), hipMemcpyDeviceToHost, st));
    }
    if (io.counts) {
      std::vector<uint32_t> c32(nq);
      VI_HIP(hipMemcpyAsync(c32.data(), ws.counts.p, nq * 4, hipMemcpyDeviceToHost, st));
      VI_HIP(hipStreamSynchronize(st));
      for (uint64_t i = 0; i < nq; ++i) io.counts[i] = c32[i];
    }
  }
  VI_HIP(hipStreamSynchronize(st));
  if (timing) (void)hipEventElapsedTime(&stt.ms_scan, ix.cur().ev[2], ix.cur().ev[3]);
  if (timing == 1) {
    (void)hipEventElapsedTime(&stt.ms_coarse, ix.cur().ev[0], ix.cur().ev[1]);
    (void)hipEventElapsedTime(&stt.ms_group, ix.cur().ev[1], ix.cur().ev[2]);
    (void)hipEventElapsedTime(&stt.ms_merge, ix.cur().ev[3], ix.cur().ev[4]);
    (void)hipEventElapsedTime(&stt.ms_total, ix.cur().ev[0], ix.cur().ev[4]);
  }
  return VI_OK;
}

// Counting sort of nq*P (query, probe) pairs by list for the generic path (the fast path folds
// the histogram into coarse_merge_kernel).  Fills ws.{cnt,seg_start,item_start,segrun_start,pairs}.
bool grouping_fuses_query_offsets(const DeviceIndex &) { return true; }

vi_status launch_grouping(const DeviceIndex &ix, const uint32_t *probes, uint64_t nq, uint32_t P, int qg, uint32_t segb0,
                          uint64_t hstats[14], hipStream_t st, bool histogram_done, const uint32_t *qtot, uint32_t *qoff,
                          const uint32_t *pair_rank) {
  SearchWorkspace &ws = ix.cur().ws;
  const uint64_t nlists = ix.nlists;
  const uint32_t total = (uint32_t)(nq * P);
  VI_TRY(ws.cnt.reserve(2 * subbin_words(nlists)));
  VI_TRY(ws.seg_start.reserve(nlists + 1));
  VI_TRY(ws.item_start.reserve(nlists + 1));
  VI_TRY(ws.segrun_start.reserve(nlists + 1));
  VI_TRY(ws.pairs.reserve(total));
  VI_TRY(ws.pair_pos.reserve(total));
  VI_TRY(ws.tile_start.reserve(nlists + 1));
  VI_TRY(ws.stats.reserve(160));
  // (stats[0..5] and [12] are reset by list_totals_kernel; [6] .. [11] belong to the MFMA path's select)
  if (!histogram_done) {  // the coarse step of the fast paths leaves the histogram behind
    VI_HIP(hipMemsetAsync(ws.cnt.p, 0, subbin_words(nlists) * sizeof(uint32_t), st));
    hipLaunchKernelGGL(histogram_kernel, dim3((total + 255) / 256), dim3(256), 0, st, probes, ix.list_len.p,
                       (uint32_t)nlists, total, P, ws.cnt.p);
  }
  VI_TRY(launch_group_scan(ix, (uint32_t)qg, segb0, ws.tile_start.p, st, true, qtot, (uint32_t)nq, qoff));
  // the host waits for the counts (grid size, scratch) while the scatter runs
  // (into page-locked memory: a copy to the caller's stack array is staged by the runtime and costs a few microseconds
  // more on the one synchronisation point of the pipeline)
  if (!ws.hstats_pinned) VI_HIP(hipHostMalloc((void **)&ws.hstats_pinned, 16 * sizeof(uint64_t)));
  VI_HIP(hipMemcpyAsync(ws.hstats_pinned, ws.stats.p, 14 * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
  VI_HIP(hipEventRecord(ix.cur().ev[5], st));
  if (pair_rank && histogram_done)
    hipLaunchKernelGGL(group_scatter_ranked_kernel, dim3((total + 255) / 256), dim3(256), 0, st, probes, ix.list_len.p,
                       (uint32_t)nlists, P, ws.cnt.p + subbin_words(nlists), pair_rank, ws.pairs.p, total, ws.seg_start.p, ws.pair_pos.p);
  else
    hipLaunchKernelGGL(group_scatter_kernel, dim3((total + 255) / 256), dim3(256), 0, st, probes, ix.list_len.p,
                       (uint32_t)nlists, P, ws.cnt.p + subbin_words(nlists), ws.pairs.p, total, ws.seg_start.p, ws.pair_pos.p);
  VI_HIP(hipGetLastError());
  VI_HIP(hipEventSynchronize(ix.cur().ev[5]));
  std::memcpy(hstats, ws.hstats_pinned, 14 * sizeof(uint64_t));
  return VI_OK;
}

vi_status merge_partials_device(int device, uint64_t nq, uint64_t k, uint32_t parts, const float *D_parts,
                                const int64_t *I_parts, const uint64_t *tie_parts, float *D_out, int64_t *I_out) {
  if (parts == 0 || parts > kWave) return fail(VI_ERR_INVALID_INPUT, "parts must be 1..64");
  if (nq == 0 || k == 0) return VI_OK;
  VI_HIP(hipSetDevice(device));
  hipLaunchKernelGGL(merge_partials_kernel, dim3((uint32_t)((nq + kWavesPerBlock - 1) / kWavesPerBlock)),
                     dim3(kBlockThreads), 0, 0, nq, (uint32_t)k, parts, D_parts, I_parts, tie_parts, (size_t)(nq * k),
                     (size_t)(nq * k), D_out, I_out);
  VI_HIP(hipGetLastError());
  VI_HIP(hipStreamSynchronize(0));
  return VI_OK;
}

// parts packed one after the other, each [D f32 nq*k | pad to 8 B | I i64 nq*k | tie u64 nq*k]: what ONE all-gather of a
// rank's packed results produces
vi_status merge_partials_packed_device(int device, uint64_t nq, uint64_t k, uint32_t parts, const void *packed,
                                       float *D_out, int64_t *I_out) {
  if (parts == 0 || parts > kWave) return fail(VI_ERR_INVALID_INPUT, "parts must be 1..64");
  if (nq == 0 || k == 0) return VI_OK;
  VI_HIP(hipSetDevice(device));
  const size_t offI = (nq * k * 4 + 7) / 8 * 8, offT = offI + nq * k * 8, stride = offT + nq * k * 8;
  const uint8_t *b = static_cast<const uint8_t *>(packed);
  hipLaunchKernelGGL(merge_partials_kernel, dim3((uint32_t)((nq + kWavesPerBlock - 1) / kWavesPerBlock)),
                     dim3(kBlockThreads), 0, 0, nq, (uint32_t)k, parts, reinterpret_cast<const float *>(b),
                     reinterpret_cast<const int64_t *>(b + offI), reinterpret_cast<const uint64_t *>(b + offT),
                     stride / 4, stride / 8, D_out, I_out);
  VI_HIP(hipGetLastError());
  VI_HIP(hipStreamSynchronize(0));
  return VI_OK;
}

vi_status l2sq_pairs_device(const float *a, const float *b, uint64_t n, uint32_t d, int order, float *out) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(VI_ERR_DEVICE, "no HIP device visible");
  if (n == 0) return VI_OK;
  DevBuf<float> da, db, dout;
  VI_TRY(da.reserve(n * d));
  VI_TRY(db.reserve(n * d));
  VI_TRY(dout.reserve(n));
  VI_HIP(hipMemcpy(da.p, a, n * d * 4, hipMemcpyHostToDevice));
  VI_HIP(hipMemcpy(db.p, b, n * d * 4, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(l2sq_pairs_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, 0, da.p, db.p, n, d, order,
                     dout.p);
  VI_HIP(hipGetLastError());
  VI_HIP(hipMemcpy(out, dout.p, n * 4, hipMemcpyDeviceToHost));
  return VI_OK;
}

}  // namespace vi
