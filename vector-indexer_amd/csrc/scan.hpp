// scan.hpp — launchers of the exact-order scan / select / repack kernels shared by the
// search path (search_kernels.hip) and the k-means path (kmeans.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "common.hpp"

namespace vi {

// quads (float4) per vector in the block layout: dims are zero-padded to a multiple of 16 so that
// the scan kernel always consumes whole groups of 4 quads
inline uint32_t layout_dq(uint32_t dim) { return ((dim + 15) / 16) * 4; }

// (query, probe) pairs are counted / scattered per (list, query & (kSubBins - 1)): hot lists are probed by thousands of
// queries of a batch and a single counter per list would serialise their atomics
constexpr uint32_t kSubBins = 32;
// counter of (list l, sub-bin s).  Sub-bin major, every sub-bin's row on cache lines of its own: the 32 counters of a
// hot list must not share a line — atomics on one line execute one after the other at its L2 channel (the hottest lists
// of a batch are probed by every query: 10 000 atomics on one line were 0.15 ms of the coarse select)
__host__ __device__ inline uint32_t subbin_stride(uint32_t nlists) { return (nlists + 31u) & ~31u; }
__host__ __device__ inline uint32_t subbin_index(uint32_t l, uint32_t s, uint32_t nlists) { return s * subbin_stride(nlists) + l; }
inline uint64_t subbin_words(uint64_t nlists) { return (uint64_t)kSubBins * subbin_stride((uint32_t)nlists); }

struct ScanArgs {
  const float4 *blocks;   // lane-interleaved blocks (device_index.hpp)
  uint32_t dq, dim;
  const float *Q;         // nq x dim, row-major, device
  uint32_t nq;
  uint32_t K;             // entries per output run (<= 64)
  float *run_dist;        // [slots][K]
  uint32_t *run_pos;
  // COARSE: one table of nvec vectors split into S ranges of bps blocks; slot = q*S + split
  uint32_t nvec, S, bps;
  // LISTS: items derived from the grouping arrays; slot = pairs[...] = q*P + rank
  const uint32_t *first_block, *list_len, *item_start, *seg_start, *pairs;
  uint32_t nlists, P;
  // long lists are cut into <= 64 segments of >= segb0 blocks so that no work item is much
  // longer than the others; segment runs of one (query, list) pair go to seg_run_* at
  // (segrun_start[list] + pair_in_list * nseg + seg) and are merged by seg_merge_kernel
  uint32_t segb0;
  uint32_t max_blocks;  // 0 = whole list; else only the first max_blocks blocks are scanned (bound sampling)
  // dump mode (generic_search.hip): instead of selecting, write one key per candidate,
  // (dist bits << 32) | index, at dump_keys[row * dump_row + index]; COARSE: row = query,
  // index = centroid; LISTS: row = slot / P, index = dump_off[slot] + position in list
  uint64_t *dump_keys;
  uint64_t dump_row;
  const uint32_t *dump_off;
  const uint32_t *segrun_start;
  float *seg_run_dist;
  uint32_t *seg_run_pos;
};

// segmentation rule shared by the grouping, scan and segment-merge kernels
__host__ __device__ inline uint32_t list_segments(uint32_t len, uint32_t segb0, uint32_t *segb) {
  const uint32_t nblk = (len + 63) / 64;
  uint32_t sb = (nblk + 63) / 64;
  if (sb < segb0) sb = segb0;
  *segb = sb;
  // (sb is the configured segment size — a power of two — for every list below 64 * segb0 blocks: a shift; a 32-bit division
  //  by a run-time value is ~35 instructions on this GPU, and this sits in the per-probe code of both select kernels)
  if ((sb & (sb - 1u)) == 0u) return (nblk + sb - 1u) >> (uint32_t)__builtin_ctz(sb);
  return nblk == 0 ? 0u : (nblk + sb - 1) / sb;
}

// i / P for a probe count that is a power of two more often than not
__host__ __device__ inline uint32_t div_probes(uint32_t i, uint32_t P) {
  if ((P & (P - 1u)) == 0u) return i >> (uint32_t)__builtin_ctz(P);
  return i / P;
}

// ceil(c / qg) for a group width that is a power of two in practice (32 / 128 / 256)
__host__ __device__ inline uint32_t group_chunks(uint32_t c, uint32_t qg) {
  if ((qg & (qg - 1u)) == 0u) return (c + qg - 1u) >> (uint32_t)__builtin_ctz(qg);
  return (c + qg - 1u) / qg;
}

// pair records one list segment of segb blocks reserves per query group (every segment of a list reserves the same
// number): a record holds four sub-block minima.  The streaming rank kernel fills them wave by wave — record
// (i >> 2) * 4 + w, component i & 3 = 32-vector tile 4 i + w of the segment — the block-synchronous kernels in pair order
// (record p = blocks 2p, 2p + 1); 4 * ceil(segb / 8) covers both.
__host__ __device__ inline uint32_t seg_records(uint32_t segb) { return 4u * ((segb + 7u) / 8u); }

// query-group width for a unit (table or list) probed by `avg_queries_per_unit` queries
int pick_qg(uint32_t dq, double avg_queries_per_unit, int order);
vi_status launch_scan(const ScanArgs &a, int qg, int order, bool coarse, uint32_t nitems_upper, hipStream_t st);
// COARSE-mode split of a table of nblk blocks for nq queries: returns S and sets *bps
uint32_t coarse_splits(uint64_t nq, int qg, uint32_t nblk, uint32_t *bps);

// Rows of a row-major device matrix -> lane-interleaved blocks.  row_of_slot[s] = source row
// of destination slot s (slot = 64*block + lane) or ~0u for a pad lane.  ids_out (optional)
// receives id_of_row[row] (or the row index itself when id_of_row is null).
vi_status launch_repack_rows(const float *src, uint32_t dim, uint32_t dq, const uint32_t *row_of_slot,
                             uint64_t nslots, const uint64_t *id_of_row, float *blocks, uint64_t *ids_out,
                             hipStream_t st);

}  // namespace vi
