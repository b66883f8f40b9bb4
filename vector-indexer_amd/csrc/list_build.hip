// list_build.hip — the list assembly of IvfIndex::fit_with_paths (src/ivf_index.rs:88-171) on the GPU.
//
// The reference buckets cloned `Vector`s into k `IVFList`s in ascending internal id (:94-101), drops the empty lists,
// groups the rest by super-centroid label into shards and writes one `shard_<id>.bin` per shard (src/shards.rs:68-177).
// Here the points never leave HBM between k-means and search:
//
//   labels (device, from the final assignment)
//     -> ids grouped by list, ascending inside a list: a stable radix sort on the label (8 bits per pass, two passes up
//        to 65 536 lists) — no atomics decide a position, so the order is the reference's, run after run
//     -> list offsets by binary search of the sorted labels
//     -> the lane-interleaved blocks + bf16 images of the resident index straight from X (device_index_from_order):
//        nothing is re-read from the shard files the build has just written
//     -> shard files: one kernel per shard lays the records (24 B meta + D f32 + pad) out in file order in a staging
//        image, one copy to pinned host memory, one write(2).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <vector>

#include "device_index.hpp"
#include "scan.hpp"
#include "shards.hpp"

namespace vi {

// search_kernels.hip
vi_status init_device_index_pub(DeviceIndex *ix, int device, uint32_t dim, uint64_t nlists);

namespace {

constexpr int kWave = 64;

// ---- stable grouping of ids by label: least-significant-digit radix sort, 8 bits per pass -------------------------
// A pass reads (label, id) pairs in their current order and writes them grouped by one digit of the label, keeping the
// order inside a digit — after the last pass the ids are grouped by label, ascending inside a label (the first pass reads
// the ids as positions, i.e. ascending).  No atomics decide a position: the result is the reference's ascending-id lists
// (src/ivf_index.rs:94-101) and the ascending member order of update_centroids_parallel (src/kmeans.rs:693-697), run
// after run.  Work per pass: 8 B read + 8 B written per element and a 256 x tiles histogram — HBM bound.
constexpr int kRadixThreads = 256, kRadixItems = 16, kRadixTile = kRadixThreads * kRadixItems;  // 4096 elements per tile

// element e of a tile belongs to wave e / 1024, item (e % 1024) / 64, lane e % 64: an item is one coalesced load
__device__ __forceinline__ uint64_t radix_elem(uint64_t tile, int wave, int item, int lane) {
  return tile * kRadixTile + (uint64_t)wave * (kRadixItems * kWave) + (uint64_t)item * kWave + lane;
}

// counts[digit * ntiles + tile] = elements of the tile with that digit
__global__ void __launch_bounds__(kRadixThreads) radix_hist_kernel(const uint32_t *keys, uint64_t n, uint32_t shift,
                                                                   uint32_t ntiles, uint32_t *counts) {
  __shared__ uint32_t hist[256];
  hist[threadIdx.x] = 0;
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int it = 0; it < kRadixItems; ++it) {
    const uint64_t e = radix_elem(blockIdx.x, wave, it, lane);
    if (e < n) atomicAdd(&hist[(keys[e] >> shift) & 255u], 1u);
  }
  __syncthreads();
  counts[(size_t)threadIdx.x * ntiles + blockIdx.x] = hist[threadIdx.x];
}

// per digit: exclusive scan of its row of tile counts in place, digit_total[digit] = the row's sum
__global__ void __launch_bounds__(256) radix_scan_rows_kernel(uint32_t *counts, uint32_t ntiles, uint32_t *digit_total) {
  __shared__ uint32_t wsum[4];
  uint32_t *row = counts + (size_t)blockIdx.x * ntiles;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  uint32_t carry = 0;
  for (uint32_t t0 = 0; t0 < ntiles; t0 += 256) {
    const uint32_t t = t0 + threadIdx.x;
    const uint32_t v = t < ntiles ? row[t] : 0u;
    uint32_t inc = v;  // inclusive scan inside the wave
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t up = __shfl_up(inc, o, 64);
      if (lane >= o) inc += up;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    uint32_t before = 0, all = 0;
    for (int w = 0; w < 4; ++w) { if (w < wave) before += wsum[w]; all += wsum[w]; }
    if (t < ntiles) row[t] = carry + before + inc - v;
    carry += all;
    __syncthreads();
  }
  if (threadIdx.x == 0) digit_total[blockIdx.x] = carry;
}

// digit_base[d] = elements with a smaller digit (256 threads, one workgroup)
__global__ void __launch_bounds__(256) radix_scan_digits_kernel(const uint32_t *digit_total, uint32_t *digit_base) {
  __shared__ uint32_t tot[256];
  tot[threadIdx.x] = digit_total[threadIdx.x];
  __syncthreads();
  uint32_t b = 0;
  for (uint32_t d = 0; d < threadIdx.x; ++d) b += tot[d];
  digit_base[threadIdx.x] = b;
}

// ids_in == nullptr: the id of element e is e
__global__ void __launch_bounds__(kRadixThreads) radix_scatter_kernel(const uint32_t *keys_in, const uint32_t *ids_in, uint64_t n,
                                                                      uint32_t shift, uint32_t ntiles, const uint32_t *counts,
                                                                      const uint32_t *digit_base, uint32_t *keys_out,
                                                                      uint32_t *ids_out) {
  __shared__ uint32_t cnt[4][256];   // per wave: elements of each digit seen so far (then: its first output position)
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int w = 0; w < 4; ++w) cnt[w][threadIdx.x] = 0;
  __syncthreads();
  uint32_t key[kRadixItems], rank[kRadixItems];
  const uint64_t lt = lane ? (~0ull >> (64 - lane)) : 0ull;
#pragma unroll
  for (int it = 0; it < kRadixItems; ++it) {
    const uint64_t e = radix_elem(blockIdx.x, wave, it, lane);
    const bool valid = e < n;
    key[it] = valid ? keys_in[e] : 0u;
    const uint32_t digit = (key[it] >> shift) & 255u;
    uint64_t peers = __ballot(valid);  // lanes of this item with the same digit
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const bool bit = (digit >> b) & 1u;
      const uint64_t m = __ballot(bit);
      peers &= bit ? m : ~m;
    }
    const uint32_t before = (uint32_t)__popcll(peers & lt);
    const uint32_t prior = cnt[wave][digit];                 // (a wave's LDS accesses execute in program order)
    if (valid && before == 0) cnt[wave][digit] = prior + (uint32_t)__popcll(peers);
    rank[it] = prior + before;
  }
  __syncthreads();
  {  // thread = digit: where each wave's elements of this digit start in the output
    const uint32_t dgt = threadIdx.x;
    uint32_t pos = digit_base[dgt] + counts[(size_t)dgt * ntiles + blockIdx.x];
    for (int w = 0; w < 4; ++w) { const uint32_t c = cnt[w][dgt]; cnt[w][dgt] = pos; pos += c; }
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < kRadixItems; ++it) {
    const uint64_t e = radix_elem(blockIdx.x, wave, it, lane);
    if (e < n) {
      const uint32_t pos = cnt[wave][(key[it] >> shift) & 255u] + rank[it];
      keys_out[pos] = key[it];
      ids_out[pos] = ids_in ? ids_in[e] : (uint32_t)e;
    }
  }
}

// off[c] = first sorted position whose label is >= c (c = 0..k)
__global__ void offsets_kernel(const uint32_t *sorted_labels, uint64_t n, uint32_t k, uint32_t *off) {
  const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c > k) return;
  uint64_t lo = 0, hi = n;  // first i in [0, n] with sorted_labels[i] >= c
  while (lo < hi) {
    const uint64_t mid = (lo + hi) >> 1;
    if (sorted_labels[mid] < c) lo = mid + 1; else hi = mid;
  }
  off[c] = (uint32_t)lo;
}

// one workgroup per kept list: row_of_slot of its blocks (pad lanes = kNoPos)
__global__ void slot_rows_kernel(const uint32_t *order, const uint64_t *src_off, const uint32_t *first_block,
                                 const uint32_t *list_len, uint32_t nlists, uint32_t *row_of_slot) {
  const uint32_t l = blockIdx.x;
  if (l >= nlists) return;
  const uint32_t len = list_len[l], padded = (len + 63u) & ~63u;
  const uint64_t so = src_off[l];
  uint32_t *dst = row_of_slot + (size_t)first_block[l] * kWave;
  for (uint32_t j = threadIdx.x; j < padded; j += blockDim.x) dst[j] = j < len ? order[so + j] : kNoPos;
}

// records of one shard in file order.  One wave per record: 32-bit words of {id, external_id, timestamp, D x f32, pad}
struct ExportArgs {
  const float *X;
  const uint32_t *order;         // ids grouped by list
  const uint64_t *ext_ids;       // per point, or null: the id itself (bindings/python/src/lib.rs:236-243)
  const uint64_t *timestamps;    // per point (0 => now), or null: now (vector_store.rs:36-40)
  uint64_t now;
  const uint64_t *rec_src;       // per list of the shard: offset into order
  const uint64_t *rec_dst;       // per list: byte offset of its first record inside the staging image
  const uint32_t *rec_len;       // per list
  const uint32_t *rec_first;     // per list: index of its first record among the shard's records (exclusive scan of rec_len)
  uint32_t nlists, dim, stride;  // stride = bytes per record
  uint64_t nrec;                 // records of the shard
  uint8_t *image;
};

__global__ void __launch_bounds__(256) export_records_kernel(ExportArgs a) {
  const int lane = threadIdx.x & 63;
  const uint64_t rec = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (rec >= a.nrec) return;
  uint32_t lo = 0, hi = a.nlists;  // largest l with rec_first[l] <= rec (lists of the shard are non-empty)
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (a.rec_first[mid] <= rec) lo = mid; else hi = mid;
  }
  const uint32_t j = (uint32_t)(rec - a.rec_first[lo]);
  const uint32_t id = a.order[a.rec_src[lo] + j];
  uint32_t *dst = reinterpret_cast<uint32_t *>(a.image + a.rec_dst[lo] + (uint64_t)j * a.stride);
  const uint64_t ext = a.ext_ids ? a.ext_ids[id] : (uint64_t)id;
  uint64_t ts = a.timestamps ? a.timestamps[id] : 0ull;
  if (ts == 0) ts = a.now;
  const uint32_t *src = reinterpret_cast<const uint32_t *>(a.X + (size_t)id * a.dim);
  const uint32_t words = a.stride / 4;
  for (uint32_t w = lane; w < words; w += kWave) {
    uint32_t v = 0;
    if (w == 0) v = id;                     // internal id = position (vector_store.rs:33), high word 0
    else if (w == 2) v = (uint32_t)ext;
    else if (w == 3) v = (uint32_t)(ext >> 32);
    else if (w == 4) v = (uint32_t)ts;
    else if (w == 5) v = (uint32_t)(ts >> 32);
    else if (w >= 6 && w < 6 + a.dim) v = src[w - 6];
    dst[w] = v;
  }
}

template <typename T>
vi_status upload(DevBuf<T> &buf, const T *host, size_t n, hipStream_t st) {
  VI_TRY(buf.reserve(std::max<size_t>(n, 1)));
  if (n) VI_HIP(hipMemcpyAsync(buf.p, host, n * sizeof(T), hipMemcpyHostToDevice, st));
  return VI_OK;
}

}  // namespace

// ids 0..n-1 grouped by label (ascending id inside a label): order (device, n u32), seg (device, k+1 u32: list c is
// order[seg[c] .. seg[c+1])) and, if asked for, the same offsets on the host.  Labels must be < k.
vi_status group_ids_by_label_device(const uint32_t *labels_dev, uint64_t n, uint64_t k, DevBuf<uint32_t> &order,
                                    DevBuf<uint32_t> &seg, std::vector<uint64_t> *off_host, hipStream_t st,
                                    DevBuf<uint32_t> *scratch_keep) {
  if (n > 0xFFFFFFFEull || k > 0xFFFFFFFEull) return fail(VI_ERR_INVALID_INPUT, "more than 2^32 - 2 points or lists");
  VI_TRY(order.reserve(std::max<uint64_t>(n, 1)));
  VI_TRY(seg.reserve(k + 1));
  if (n == 0) {  // a rank that owns no points: every list empty
    VI_HIP(hipMemsetAsync(seg.p, 0, (k + 1) * 4, st));
    if (off_host) off_host->assign(k + 1, 0);
    VI_HIP(hipStreamSynchronize(st));
    return VI_OK;
  }
  uint32_t bits = 1;
  while (bits < 32 && ((k - 1) >> bits) != 0) ++bits;
  const uint32_t passes = (bits + 7) / 8;
  const uint32_t ntiles = (uint32_t)((n + kRadixTile - 1) / kRadixTile);
  DevBuf<uint32_t> keys[2], ids_tmp, counts, digit_total, digit_base;
  VI_TRY(keys[0].reserve(n));
  if (passes > 1) { VI_TRY(keys[1].reserve(n)); VI_TRY(ids_tmp.reserve(n)); }
  VI_TRY(counts.reserve((size_t)256 * ntiles));
  VI_TRY(digit_total.reserve(256));
  VI_TRY(digit_base.reserve(256));
  // the ids ping-pong between ids_tmp and order such that the LAST pass writes into order
  const uint32_t *kin = labels_dev, *iin = nullptr;
  for (uint32_t p = 0; p < passes; ++p) {
    uint32_t *kout = keys[p & 1].p;
    uint32_t *iout = ((passes - 1 - p) & 1) ? ids_tmp.p : order.p;
    hipLaunchKernelGGL(radix_hist_kernel, dim3(ntiles), dim3(kRadixThreads), 0, st, kin, n, 8 * p, ntiles, counts.p);
    hipLaunchKernelGGL(radix_scan_rows_kernel, dim3(256), dim3(256), 0, st, counts.p, ntiles, digit_total.p);
    hipLaunchKernelGGL(radix_scan_digits_kernel, dim3(1), dim3(256), 0, st, digit_total.p, digit_base.p);
    hipLaunchKernelGGL(radix_scatter_kernel, dim3(ntiles), dim3(kRadixThreads), 0, st, kin, iin, n, 8 * p, ntiles, counts.p,
                       digit_base.p, kout, iout);
    VI_HIP(hipGetLastError());
    kin = kout;
    iin = iout;
  }
  hipLaunchKernelGGL(offsets_kernel, dim3((uint32_t)((k + 1 + 255) / 256)), dim3(256), 0, st, kin, n, (uint32_t)k, seg.p);
  VI_HIP(hipGetLastError());
  if (off_host) {
    std::vector<uint32_t> o32(k + 1);
    VI_HIP(hipMemcpyAsync(o32.data(), seg.p, (k + 1) * 4, hipMemcpyDeviceToHost, st));
    VI_HIP(hipStreamSynchronize(st));
    off_host->assign(o32.begin(), o32.end());
  } else {
    VI_HIP(hipStreamSynchronize(st));  // the scratch buffers above are freed on return
  }
  return VI_OK;
}

// The resident index from device data: table_host = the kept centroids (nlists x dim), list l = rows
// order[src_off[l] .. src_off[l] + len[l]) of X_dev, ids_dev (optional) = external id per row.
vi_status device_index_from_order(int device, uint32_t dim, const float *table_host, uint64_t nlists, const float *X_dev,
                                  const uint32_t *order_dev, const std::vector<uint64_t> &src_off,
                                  const std::vector<uint32_t> &len, const std::vector<uint32_t> &list_shard,
                                  const uint64_t *ids_dev, DeviceIndex *ix) {
  VI_TRY(init_device_index_pub(ix, device, dim, nlists));
  ix->order = VI_ORDER_SCALAR;
  const uint32_t dq = ix->dq;
  hipStream_t st = ix->stream;
  {  // coarse table
    const uint64_t nb = (nlists + kWave - 1) / kWave;
    ix->centroids.dq = dq;
    ix->centroids.nblocks = nb;
    VI_TRY(ix->centroids.blocks.reserve(std::max<uint64_t>(1, nb) * dq * kWave * 4));
    std::vector<uint32_t> ros(nb * kWave, kNoPos);
    for (uint64_t i = 0; i < nlists; ++i) ros[i] = (uint32_t)i;
    DevBuf<uint32_t> dros;
    DevBuf<float> dtab;
    VI_TRY(upload(dros, ros.data(), ros.size(), st));
    VI_TRY(upload(dtab, table_host, nlists * dim, st));
    VI_TRY(launch_repack_rows(dtab.p, dim, dq, dros.p, ros.size(), nullptr, ix->centroids.blocks.p, nullptr, st));
    VI_HIP(hipStreamSynchronize(st));
  }
  std::vector<uint32_t> h_first(nlists, 0);
  uint64_t total_blocks = 0, total_vec = 0, nshards = 0;
  for (uint64_t l = 0; l < nlists; ++l) {
    h_first[l] = (uint32_t)total_blocks;
    total_blocks += (len[l] + kWave - 1) / kWave;
    total_vec += len[l];
    nshards = std::max<uint64_t>(nshards, (uint64_t)list_shard[l] + 1);
  }
  if (total_blocks >= 0xFFFFFFFFull / kWave) return fail(VI_ERR_OTHER, "index too large for 32-bit slot ids");
  ix->nvec_resident = total_vec;
  ix->nshards = nshards;
  ix->lists.dq = dq;
  ix->lists.nblocks = total_blocks;
  VI_TRY(ix->lists.blocks.reserve(std::max<uint64_t>(1, total_blocks) * dq * kWave * 4));
  VI_TRY(ix->ext_ids.reserve(std::max<uint64_t>(1, total_blocks) * kWave));
  VI_TRY(ix->list_first_block.reserve(std::max<uint64_t>(1, nlists)));
  VI_TRY(ix->list_len.reserve(std::max<uint64_t>(1, nlists)));
  VI_TRY(ix->list_shard.reserve(std::max<uint64_t>(1, nlists)));
  if (nlists) {
    VI_HIP(hipMemcpyAsync(ix->list_first_block.p, h_first.data(), nlists * 4, hipMemcpyHostToDevice, st));
    VI_HIP(hipMemcpyAsync(ix->list_len.p, len.data(), nlists * 4, hipMemcpyHostToDevice, st));
    VI_HIP(hipMemcpyAsync(ix->list_shard.p, list_shard.data(), nlists * 4, hipMemcpyHostToDevice, st));
    DevBuf<uint64_t> dsrc;
    DevBuf<uint32_t> dros;
    VI_TRY(upload(dsrc, src_off.data(), nlists, st));
    VI_TRY(dros.reserve(std::max<uint64_t>(1, total_blocks * kWave)));
    hipLaunchKernelGGL(slot_rows_kernel, dim3((uint32_t)nlists), dim3(256), 0, st, order_dev, dsrc.p, ix->list_first_block.p,
                       ix->list_len.p, (uint32_t)nlists, dros.p);
    VI_HIP(hipGetLastError());
    VI_TRY(launch_repack_rows(X_dev, dim, dq, dros.p, total_blocks * kWave, ids_dev, ix->lists.blocks.p, ix->ext_ids.p, st));
    VI_HIP(hipStreamSynchronize(st));
  }
  return compute_slot_norms(ix);
}

// One shard file from device-resident points (byte-identical to shard_save_to, shards.cpp): lists = the shard's lists in
// file order, each (centroid id, centroid vector (host), offset into order_dev, length).
vi_status shard_export_device(const std::string &shards_dir, uint64_t shard_id, uint32_t dim, const std::vector<uint64_t> &cids,
                              const float *cvecs_host, const std::vector<uint64_t> &src_off, const std::vector<uint32_t> &len,
                              const float *X_dev, const uint32_t *order_dev, const uint64_t *ext_dev, const uint64_t *ts_dev,
                              uint64_t now, ShardExportWs &ws, hipStream_t st) {
  VI_TRY(make_dirs(shards_dir));
  const std::string path = shards_dir + "/shard_" + std::to_string(shard_id) + ".bin";
  ::remove(path.c_str());  // shards.rs:73
  const uint32_t nl = (uint32_t)cids.size();
  const uint64_t vsz = 4ull * dim, pad = pad8(vsz), stride = record_stride(dim);
  const uint64_t data_off = kShardHeaderBytes + kIndexEntryBytes * (uint64_t)nl;
  std::vector<uint64_t> rec_dst(nl), blk_off(nl);
  std::vector<uint32_t> rec_first(nl);
  uint64_t cur = data_off, nrec = 0;
  for (uint32_t i = 0; i < nl; ++i) {
    blk_off[i] = cur;
    rec_dst[i] = cur + vsz + pad;
    rec_first[i] = (uint32_t)nrec;
    cur += vsz + pad + (uint64_t)len[i] * stride;
    nrec += len[i];
  }
  const uint64_t total = cur;
  if (nrec >= 0xFFFFFFFFull) return fail(VI_ERR_OTHER, "shard with more than 2^32 records");
  // staging image on the device (records only; header, index and centroid vectors are patched in on the host)
  VI_TRY(ws.image.reserve(total + 16));
  if (ws.host_cap < total) {
    if (ws.host) (void)hipHostFree(ws.host);
    ws.host = nullptr;
    ws.host_cap = 0;
    VI_HIP(hipHostMalloc((void **)&ws.host, total + 16));
    ws.host_cap = total;
  }
  // only lists with records take part in the kernel's search table
  std::vector<uint64_t> k_src, k_dst;
  std::vector<uint32_t> k_len, k_first;
  for (uint32_t i = 0; i < nl; ++i)
    if (len[i]) { k_src.push_back(src_off[i]); k_dst.push_back(rec_dst[i]); k_len.push_back(len[i]); k_first.push_back(rec_first[i]); }
  if (nrec) {
    VI_TRY(upload(ws.d_src, k_src.data(), k_src.size(), st));
    VI_TRY(upload(ws.d_dst, k_dst.data(), k_dst.size(), st));
    VI_TRY(upload(ws.d_len, k_len.data(), k_len.size(), st));
    VI_TRY(upload(ws.d_first, k_first.data(), k_first.size(), st));
    ExportArgs a{X_dev, order_dev, ext_dev, ts_dev, now, ws.d_src.p, ws.d_dst.p, ws.d_len.p, ws.d_first.p,
                 (uint32_t)k_src.size(), dim, (uint32_t)stride, nrec, ws.image.p};
    hipLaunchKernelGGL(export_records_kernel, dim3((uint32_t)((nrec + 3) / 4)), dim3(256), 0, st, a);
    VI_HIP(hipGetLastError());
    VI_HIP(hipMemcpyAsync(ws.host + data_off, ws.image.p + data_off, total - data_off, hipMemcpyDeviceToHost, st));
    VI_HIP(hipStreamSynchronize(st));  // (k_* are read by the copies above)
  }
  uint8_t *p = ws.host;
  auto put64 = [&](uint64_t off, uint64_t v) { std::memcpy(p + off, &v, 8); };
  auto put32 = [&](uint64_t off, uint32_t v) { std::memcpy(p + off, &v, 4); };
  put64(0, shard_id); put64(8, 1); put32(16, dim); put32(20, nl);
  put64(24, kShardHeaderBytes); put64(32, data_off);
  for (uint32_t i = 0; i < nl; ++i) {
    const uint64_t e = kShardHeaderBytes + (uint64_t)i * kIndexEntryBytes;
    put64(e, cids[i]); put32(e + 8, len[i]); put32(e + 12, 0);
    put64(e + 16, blk_off[i]); put64(e + 24, vsz + pad + (uint64_t)len[i] * stride);
    std::memcpy(p + blk_off[i], cvecs_host + (uint64_t)i * dim, vsz);
    if (pad) std::memset(p + blk_off[i] + vsz, 0, pad);
  }
  FILE *f = fopen(path.c_str(), "wb");
  if (!f) return fail(VI_ERR_IO, "File::create(%s): %s", path.c_str(), strerror(errno));
  const bool ok = total == 0 || fwrite(p, 1, total, f) == total;
  if (fclose(f) != 0 || !ok) return fail(VI_ERR_IO, "write(%s) failed", path.c_str());
  return VI_OK;
}

ShardExportWs::~ShardExportWs() {
  if (host) (void)hipHostFree(host);
}

}  // namespace vi
