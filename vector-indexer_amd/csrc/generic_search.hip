// generic_search.hip — the any-k / any-n_probe search path.
//
// The wave-resident selection of search_kernels.hip covers k <= 64 and n_probe <= 64.  Beyond
// that (the API allows max_k = max_n_probe = 10 000, src/api.rs:40-41) the search is executed
// the way the reference literally states it (src/ivf_index.rs:205-266): compute EVERY distance,
// stable-sort, take the first n_probe / k — as device-wide passes:
//
//   scan_kernel<COARSE> in dump mode : key = (dist bits << 32 | centroid index)  for all centroids
//   bitonic sort of each query row   : == stable sort by distance (index breaks ties)
//   shard visiting order             : first appearance in the probe list, again by a key sort
//   scan_kernel<LISTS> in dump mode  : key = (dist bits << 32 | candidate index in reference order)
//   bitonic sort, first k keys       : == the reference's stable candidate sort
//
// Distances are the same exact-order f32 sums as on the fast path, so results are bit-identical
// to it wherever both apply (tests/test_search_gpu.py::test_generic_path_*).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <vector>

#include "device_index.hpp"
#include "scan.hpp"

namespace vi {
namespace {

constexpr uint32_t kChunk = 2048;          // keys sorted per workgroup in LDS (16 KiB)
constexpr uint64_t kMaxKeys = 1ull << 27;  // keys in flight per chunk of queries (1 GiB)

__device__ __forceinline__ void cmp_swap(uint64_t &x, uint64_t &y, bool asc) {
  if ((x > y) == asc) { const uint64_t t = x; x = y; y = t; }
}

// sort every kChunk-aligned chunk completely (all stages with k2 <= kChunk); L % kChunk == 0
__global__ void __launch_bounds__(1024) bitonic_local_sort_kernel(uint64_t *keys, uint32_t logL) {
  __shared__ uint64_t s[kChunk];
  const uint64_t base = (uint64_t)blockIdx.x * kChunk;
  const uint64_t in_row = base & ((1ull << logL) - 1);  // sort direction follows the index INSIDE the row
  for (uint32_t i = threadIdx.x; i < kChunk; i += 1024) s[i] = keys[base + i];
  __syncthreads();
  for (uint32_t k2 = 2; k2 <= kChunk; k2 <<= 1)
    for (uint32_t j = k2 >> 1; j > 0; j >>= 1) {
      const uint32_t i = threadIdx.x;  // kChunk/2 == 1024 pairs
      const uint32_t a = ((i & ~(j - 1)) << 1) | (i & (j - 1));
      const bool asc = (((in_row + a) & k2) == 0);
      cmp_swap(s[a], s[a + j], asc);
      __syncthreads();
    }
  for (uint32_t i = threadIdx.x; i < kChunk; i += 1024) keys[base + i] = s[i];
}

// one global compare-exchange stage (j >= kChunk); rows have length L (power of two)
__global__ void bitonic_global_step_kernel(uint64_t *keys, uint64_t npairs, uint32_t logL, uint64_t k2, uint64_t j) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= npairs) return;
  const uint64_t half = 1ull << (logL - 1);
  const uint64_t row = t >> (logL - 1), i = t & (half - 1);
  const uint64_t a = ((i & ~(j - 1)) << 1) | (i & (j - 1));
  const bool asc = ((a & k2) == 0);
  uint64_t *r = keys + (row << logL);
  uint64_t x = r[a], y = r[a + j];
  if ((x > y) == asc) { r[a] = y; r[a + j] = x; }
}

// finish a merge stage k2 > kChunk inside LDS: steps j = kChunk/2 .. 1 of every chunk
__global__ void __launch_bounds__(1024) bitonic_local_merge_kernel(uint64_t *keys, uint32_t logL, uint64_t k2) {
  __shared__ uint64_t s[kChunk];
  const uint64_t base = (uint64_t)blockIdx.x * kChunk;
  const uint64_t in_row = base & ((1ull << logL) - 1);
  for (uint32_t i = threadIdx.x; i < kChunk; i += 1024) s[i] = keys[base + i];
  __syncthreads();
  for (uint32_t j = kChunk >> 1; j > 0; j >>= 1) {
    const uint32_t i = threadIdx.x;
    const uint32_t a = ((i & ~(j - 1)) << 1) | (i & (j - 1));
    const bool asc = (((in_row + a) & k2) == 0);
    cmp_swap(s[a], s[a + j], asc);
    __syncthreads();
  }
  for (uint32_t i = threadIdx.x; i < kChunk; i += 1024) keys[base + i] = s[i];
}

// sort `nrows` rows of length L = 2^logL (L >= kChunk) ascending
vi_status sort_rows(uint64_t *keys, uint64_t nrows, uint32_t logL, hipStream_t st) {
  const uint64_t L = 1ull << logL, total = nrows * L;
  if (total == 0) return VI_OK;
  const uint32_t nchunks = (uint32_t)(total / kChunk);
  hipLaunchKernelGGL(bitonic_local_sort_kernel, dim3(nchunks), dim3(1024), 0, st, keys, logL);
  for (uint64_t k2 = 2ull * kChunk; k2 <= L; k2 <<= 1) {
    for (uint64_t j = k2 >> 1; j >= kChunk; j >>= 1) {
      const uint64_t npairs = total / 2;
      hipLaunchKernelGGL(bitonic_global_step_kernel, dim3((uint32_t)((npairs + 255) / 256)), dim3(256), 0, st, keys,
                         npairs, logL, k2, j);
    }
    hipLaunchKernelGGL(bitonic_local_merge_kernel, dim3(nchunks), dim3(1024), 0, st, keys, logL, k2);
  }
  VI_HIP(hipGetLastError());
  return VI_OK;
}

uint32_t log2_ceil_rows(uint64_t n) {  // row length exponent, at least kChunk
  uint32_t lg = 11;
  while ((1ull << lg) < n) ++lg;
  return lg;
}

__global__ void take_probes_kernel(const uint64_t *keys, uint32_t logL, uint32_t nq, uint32_t P, uint32_t *probes) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (uint64_t)nq * P) return;
  const uint64_t q = t / P, r = t % P;
  probes[t] = (uint32_t)keys[(q << logL) + r];
}

// key = (first appearance of the probe's shard in the probe list, probe rank)
__global__ void shard_order_keys_kernel(const uint32_t *probes, const uint32_t *list_shard, uint32_t P, uint32_t logLp,
                                        uint32_t nshards, uint64_t *keys) {
  extern __shared__ uint32_t fa[];
  const uint32_t q = blockIdx.x;
  for (uint32_t s = threadIdx.x; s < nshards; s += blockDim.x) fa[s] = kNoPos;
  __syncthreads();
  // (probes < nlists: P <= nlists, so a sorted row starts with P real keys; a shard id out of range is skipped)
  for (uint32_t r = threadIdx.x; r < P; r += blockDim.x) {
    const uint32_t sh = list_shard[probes[(size_t)q * P + r]];
    if (sh < nshards) atomicMin(&fa[sh], r);
  }
  __syncthreads();
  const uint32_t Lp = 1u << logLp;
  for (uint32_t r = threadIdx.x; r < Lp; r += blockDim.x)
    keys[((size_t)q << logLp) + r] =
        r < P ? (((uint64_t)fa[min(list_shard[probes[(size_t)q * P + r]], nshards - 1u)] << 32) | r) : ~0ull;
}

// per query: candidate offsets of each probe in reference candidate order (sequential prefix)
__global__ void candidate_offsets_kernel(const uint64_t *order_keys, uint32_t logLp, const uint32_t *probes,
                                         const uint32_t *list_len, uint32_t nq, uint32_t P, uint32_t *gprobe,
                                         uint32_t *off_by_g, uint32_t *off_by_rank, uint32_t *gorder, uint64_t *total) {
  const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= nq) return;
  uint64_t run = 0;
  for (uint32_t g = 0; g < P; ++g) {
    const uint32_t r = (uint32_t)order_keys[((size_t)q << logLp) + g];
    gprobe[(size_t)q * P + g] = r;
    gorder[(size_t)q * P + r] = g;
    off_by_g[(size_t)q * P + g] = (uint32_t)run;
    off_by_rank[(size_t)q * P + r] = (uint32_t)run;
    run += list_len[probes[(size_t)q * P + r]];
  }
  total[q] = run;
}

// probe lists given by the caller (another rank's coarse step): the candidate-order ranks of a row's real probes must be
// distinct — a duplicate would leave a probe without its place in the key rows.  One workgroup per query, a bitmap in LDS.
__global__ void __launch_bounds__(256) validate_order_kernel(const uint32_t *probes, const uint32_t *order, uint32_t P, uint32_t *bad) {
  extern __shared__ uint32_t seen[];
  const uint32_t q = blockIdx.x, words = (P + 31u) / 32u;
  for (uint32_t w = threadIdx.x; w < words; w += blockDim.x) seen[w] = 0u;
  __syncthreads();
  for (uint32_t r = threadIdx.x; r < P; r += blockDim.x) {
    if (probes[(size_t)q * P + r] == kNoPos) continue;
    const uint32_t g = order[(size_t)q * P + r];
    if (g >= P || (atomicOr(&seen[g >> 5], 1u << (g & 31u)) >> (g & 31u)) & 1u) atomicOr(bad, 1u);
  }
}

// the arrays candidate_offsets_kernel derives from sorted shard keys, from a given order instead: gprobe must be
// preset to kNoPos (ranks no real probe holds stay empty)
__global__ void offsets_from_order_kernel(const uint32_t *probes, const uint32_t *gorder, const uint32_t *list_len, uint32_t nq, uint32_t P,
                                          uint32_t *gprobe, uint32_t *off_by_g, uint32_t *off_by_rank, uint64_t *total) {
  const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= nq) return;
  for (uint32_t r = 0; r < P; ++r)
    if (probes[(size_t)q * P + r] != kNoPos) gprobe[(size_t)q * P + gorder[(size_t)q * P + r]] = r;
  uint64_t run = 0;
  for (uint32_t g = 0; g < P; ++g) {
    const uint32_t r = gprobe[(size_t)q * P + g];
    off_by_g[(size_t)q * P + g] = (uint32_t)run;
    if (r == kNoPos) continue;
    off_by_rank[(size_t)q * P + r] = (uint32_t)run;
    run += list_len[probes[(size_t)q * P + r]];
  }
  total[q] = run;
}

struct OutArgs {
  const uint64_t *keys;
  uint32_t logL, nq, P, k;
  const uint32_t *probes, *gprobe, *off_by_g, *first_block;
  const uint64_t *total, *ext_ids;
  float *D;
  int64_t *I;
  uint64_t *tie, *slots;
  uint32_t *counts;
};

__global__ void generic_output_kernel(OutArgs a) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (uint64_t)a.nq * a.k) return;
  const uint32_t q = (uint32_t)(t / a.k), i = (uint32_t)(t % a.k);
  const uint64_t tot = a.total[q];
  if (i == 0 && a.counts) a.counts[q] = (uint32_t)(tot < a.k ? tot : a.k);
  if (i >= tot) {
    a.D[t] = INFINITY; a.I[t] = -1;
    if (a.tie) a.tie[t] = ~0ull;
    if (a.slots) a.slots[t] = ~0ull;
    return;
  }
  const uint64_t key = a.keys[((size_t)q << a.logL) + i];
  const uint32_t ci = (uint32_t)key;
  const uint32_t *off = a.off_by_g + (size_t)q * a.P;
  uint32_t lo = 0, hi = a.P;  // largest g with off[g] <= ci; empty lists share an offset with their successor
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (off[mid] <= ci) lo = mid; else hi = mid;
  }
  const uint32_t g = lo, pos = ci - off[g];
  const uint32_t gr = a.gprobe[(size_t)q * a.P + g];
  const uint32_t l = a.probes[(size_t)q * a.P + (gr < a.P ? gr : 0u)];
  const uint64_t gslot = (uint64_t)a.first_block[l] * 64 + pos;
  a.D[t] = __uint_as_float((uint32_t)(key >> 32));
  a.I[t] = (int64_t)a.ext_ids[gslot];
  if (a.tie) a.tie[t] = ((uint64_t)g << 32) | pos;
  if (a.slots) a.slots[t] = gslot;
}

}  // namespace

// device-wide sort of rows of 2^logL u64 keys (list_build.hip groups the point ids by list with it)
vi_status sort_rows_u64(uint64_t *keys, uint64_t nrows, uint32_t logL, hipStream_t st) { return sort_rows(keys, nrows, logL, st); }

// declared in search_kernels.hip
vi_status launch_grouping(const DeviceIndex &ix, const uint32_t *probes, uint64_t nq, uint32_t P, int qg, uint32_t segb0,
                          uint64_t hstats[14], hipStream_t st, bool histogram_done, const uint32_t *qtot = nullptr,
                          uint32_t *qoff = nullptr, const uint32_t *pair_rank = nullptr);
bool grouping_fuses_query_offsets(const DeviceIndex &ix);

// declared in search_kernels.hip
vi_status adopt_probes(const DeviceIndex &ix, uint64_t nq, uint32_t P, const uint32_t *probes_in, const uint32_t *order_in,
                       bool histogram, hipStream_t st);

namespace {

// A. probes: dump all coarse distances, sort each row, take the first P -> ws.probes
vi_status generic_coarse(const DeviceIndex &ix, const float *Qd, uint64_t nq, uint32_t P, hipStream_t st) {
  SearchWorkspace &ws = ix.cur().ws;
  const uint32_t dim = ix.dim, dq = ix.dq;
  const uint64_t nlists = ix.nlists;
  VI_TRY(ws.probes.reserve(nq * P));
  VI_TRY(ws.gorder.reserve(nq * P));
  const uint32_t logLc = log2_ceil_rows(nlists);
  const uint64_t Lc = 1ull << logLc;
  const uint64_t qc = std::max<uint64_t>(1, std::min<uint64_t>(nq, kMaxKeys / Lc));
  VI_TRY(ws.sort_keys.reserve(qc * Lc));
  const uint32_t nblk_c = (uint32_t)ix.centroids.nblocks;
  for (uint64_t q0 = 0; q0 < nq; q0 += qc) {
    const uint64_t m = std::min(qc, nq - q0);
    VI_HIP(hipMemsetAsync(ws.sort_keys.p, 0xFF, m * Lc * sizeof(uint64_t), st));
    const int qg = pick_qg(dq, (double)m, ix.order);
    uint32_t bps = 0;
    const uint32_t S = coarse_splits(m, qg, nblk_c, &bps);
    ScanArgs a{};
    a.blocks = (const float4 *)ix.centroids.blocks.p; a.dq = dq; a.dim = dim; a.Q = Qd + q0 * dim; a.nq = (uint32_t)m;
    a.K = 1; a.nvec = (uint32_t)nlists; a.S = S; a.bps = bps;
    a.dump_keys = ws.sort_keys.p; a.dump_row = Lc;
    VI_TRY(launch_scan(a, qg, ix.order, true, (uint32_t)((m + qg - 1) / qg) * S, st));
    VI_TRY(sort_rows(ws.sort_keys.p, m, logLc, st));
    hipLaunchKernelGGL(take_probes_kernel, dim3((uint32_t)((m * P + 255) / 256)), dim3(256), 0, st, ws.sort_keys.p,
                       logLc, (uint32_t)m, P, ws.probes.p + q0 * P);
    VI_HIP(hipGetLastError());
  }
  return VI_OK;
}

// B. shard visiting order -> candidate order of the probes (ws.gorder and the offset arrays), from ws.probes; or, with
// `given`, from the order the caller supplied (already in ws.gorder)
vi_status generic_candidate_order(const DeviceIndex &ix, uint64_t nq, uint32_t P, bool given, hipStream_t st) {
  SearchWorkspace &ws = ix.cur().ws;
  VI_TRY(ws.gprobe.reserve(nq * P));
  VI_TRY(ws.off_by_g.reserve(nq * P));
  VI_TRY(ws.off_by_rank.reserve(nq * P));
  VI_TRY(ws.total.reserve(nq));
  if (given) {
    VI_TRY(ws.probe_flag.reserve(1));
    VI_HIP(hipMemsetAsync(ws.probe_flag.p, 0, 4, st));
    hipLaunchKernelGGL(validate_order_kernel, dim3((uint32_t)nq), dim3(256), ((P + 31u) / 32u) * 4u, st, ws.probes.p, ws.gorder.p, P,
                       ws.probe_flag.p);
    VI_HIP(hipGetLastError());
    uint32_t bad = 0;
    VI_HIP(hipMemcpyAsync(&bad, ws.probe_flag.p, 4, hipMemcpyDeviceToHost, st));
    VI_HIP(hipStreamSynchronize(st));
    if (bad) return fail(VI_ERR_INVALID_INPUT, "probe order: the ranks of a query's probes must be distinct and below n_probe_eff");
    VI_HIP(hipMemsetAsync(ws.gprobe.p, 0xFF, nq * P * sizeof(uint32_t), st));
    VI_HIP(hipMemsetAsync(ws.off_by_rank.p, 0, nq * P * sizeof(uint32_t), st));
    hipLaunchKernelGGL(offsets_from_order_kernel, dim3((uint32_t)((nq + 63) / 64)), dim3(64), 0, st, ws.probes.p, ws.gorder.p,
                       ix.list_len.p, (uint32_t)nq, P, ws.gprobe.p, ws.off_by_g.p, ws.off_by_rank.p, ws.total.p);
    VI_HIP(hipGetLastError());
    return VI_OK;
  }
  const uint32_t logLp = log2_ceil_rows(P);
  VI_TRY(ws.order_keys.reserve(nq << logLp));
  hipLaunchKernelGGL(shard_order_keys_kernel, dim3((uint32_t)nq), dim3(256), std::max<uint64_t>(1, ix.nshards) * 4, st,
                     ws.probes.p, ix.list_shard.p, P, logLp, (uint32_t)ix.nshards, ws.order_keys.p);
  VI_HIP(hipGetLastError());
  VI_TRY(sort_rows(ws.order_keys.p, nq, logLp, st));
  hipLaunchKernelGGL(candidate_offsets_kernel, dim3((uint32_t)((nq + 63) / 64)), dim3(64), 0, st, ws.order_keys.p, logLp,
                     ws.probes.p, ix.list_len.p, (uint32_t)nq, P, ws.gprobe.p, ws.off_by_g.p, ws.off_by_rank.p,
                     ws.gorder.p, ws.total.p);
  VI_HIP(hipGetLastError());
  return VI_OK;
}

}  // namespace

// the coarse step alone for any n_probe (probe export of the multi-GPU protocol): ws.probes / ws.gorder
vi_status generic_probe_export(const DeviceIndex &ix, const float *Qd, uint64_t nq, uint32_t P, hipStream_t st) {
  if (ix.nshards > 24576) return fail(VI_ERR_OTHER, "generic path supports at most 24576 shards");
  VI_TRY(generic_coarse(ix, Qd, nq, P, st));
  return generic_candidate_order(ix, nq, P, false, st);
}

vi_status device_index_search_generic(const DeviceIndex &ix, const float *Qd, uint64_t nq, uint64_t k, uint32_t P,
                                      float *Dd, int64_t *Id, uint64_t *Td, uint64_t *slots, uint32_t *counts,
                                      hipStream_t st, const uint32_t *probes_in, const uint32_t *order_in) {
  SearchWorkspace &ws = ix.cur().ws;
  const uint32_t dim = ix.dim, dq = ix.dq;
  const uint64_t nlists = ix.nlists;
  if (ix.nshards > 24576) return fail(VI_ERR_OTHER, "generic path supports at most 24576 shards");
  vi_search_stats &stt = ix.cur().stats;
  if (probes_in) {  // probe lists computed elsewhere: validated, then placed by the given order
    VI_TRY(adopt_probes(ix, nq, P, probes_in, order_in, false, st));
    VI_TRY(generic_candidate_order(ix, nq, P, true, st));
  } else {
    VI_TRY(generic_coarse(ix, Qd, nq, P, st));
    VI_TRY(generic_candidate_order(ix, nq, P, false, st));
  }
  std::vector<uint64_t> h_total(nq);
  VI_HIP(hipMemcpyAsync(h_total.data(), ws.total.p, nq * 8, hipMemcpyDeviceToHost, st));
  VI_HIP(hipStreamSynchronize(st));
  for (uint64_t q = 0; q < nq; ++q) {
    if (h_total[q] > 0xFFFFFFFFull) return fail(VI_ERR_OTHER, "more than 2^32 candidates for one query");
    stt.scanned_vectors += h_total[q];
  }

  // ---- C/D. per chunk of queries: dump candidate keys, sort rows, emit the first k ----
  const uint32_t segb0 = 256;
  uint64_t q0 = 0;
  while (q0 < nq) {
    uint64_t mx = std::max<uint64_t>(h_total[q0], 1);
    uint64_t m = 1;
    while (q0 + m < nq) {  // grow the chunk while rows x row-length stays within the key budget
      const uint64_t mx2 = std::max(mx, h_total[q0 + m]);
      if ((m + 1) << log2_ceil_rows(mx2) > kMaxKeys) break;
      mx = mx2;
      ++m;
    }
    const uint32_t logL = log2_ceil_rows(std::max<uint64_t>(mx, k));
    const uint64_t L = 1ull << logL;
    VI_TRY(ws.sort_keys.reserve(m * L));
    VI_HIP(hipMemsetAsync(ws.sort_keys.p, 0xFF, m * L * sizeof(uint64_t), st));
    const double avg_q_per_list = (double)m * P / (double)std::max<uint64_t>(1, nlists);
    const int qg = pick_qg(dq, avg_q_per_list, ix.order);
    uint64_t hstats[14];
    VI_TRY(launch_grouping(ix, ws.probes.p + q0 * P, m, P, qg, segb0, hstats, st, false));
    stt.scan_items += hstats[1];
    ScanArgs a{};
    a.blocks = (const float4 *)ix.lists.blocks.p; a.dq = dq; a.dim = dim; a.Q = Qd + q0 * dim; a.nq = (uint32_t)m;
    a.K = 1;
    a.first_block = ix.list_first_block.p; a.list_len = ix.list_len.p; a.item_start = ws.item_start.p;
    a.seg_start = ws.seg_start.p; a.pairs = ws.pairs.p; a.nlists = (uint32_t)nlists; a.P = P;
    a.segb0 = segb0; a.segrun_start = ws.segrun_start.p;
    a.dump_keys = ws.sort_keys.p; a.dump_row = L; a.dump_off = ws.off_by_rank.p + q0 * P;
    VI_TRY(launch_scan(a, qg, ix.order, false, (uint32_t)hstats[1], st));
    VI_TRY(sort_rows(ws.sort_keys.p, m, logL, st));
    OutArgs o{ws.sort_keys.p, logL, (uint32_t)m, P, (uint32_t)k, ws.probes.p + q0 * P, ws.gprobe.p + q0 * P,
              ws.off_by_g.p + q0 * P, ix.list_first_block.p, ws.total.p + q0, ix.ext_ids.p, Dd + q0 * k, Id + q0 * k,
              Td ? Td + q0 * k : nullptr, slots ? slots + q0 * k : nullptr, counts ? counts + q0 : nullptr};
    hipLaunchKernelGGL(generic_output_kernel, dim3((uint32_t)((m * k + 255) / 256)), dim3(256), 0, st, o);
    VI_HIP(hipGetLastError());
    VI_HIP(hipStreamSynchronize(st));
    q0 += m;
  }
  return VI_OK;
}

}  // namespace vi
