// rank_stream.hpp — the streaming list-rank kernel (rank_stream.hip) as filter_search.hip's pipeline launches it.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "common.hpp"

namespace vi {

struct RankStreamArgs {
  const uint4 *img;     // bf16 hi/lo image of the lists: per block nc chunks x [plane][half] x 64 columns x 16 B
  const float *xnorm;   // squared norms in image-column order
  const uint4 *qimg;    // -2 q of the batch split hi / lo: per query [plane][chunk][half] x 16 B (split_queries_kernel)
  const uint4 *sdesc;   // per work item {queries, first block, 32-vector tiles, first record tile} (item_cols_kernel)
  uint32_t nitems;
  const uint32_t *qcol;  // per (item, column of the group): the query (item_cols_kernel), ~0: none
  const uint32_t *grec;  // per (item, column): index of the group record of lane half 0, ~0: none
  uint32_t *queue;       // work counter, zero at launch
  float4 *gval;         // group records (their place words are written by group_place_kernel)
  float4 *brec;         // pair records, wave order (scan.hpp: seg_records)
  unsigned long long *prof;  // diagnostic: 8 counters of phase clocks (rank_stream.hip), or null
  uint32_t xmode;       // ablation knob (VI_FILTER_XMODE): 1 tiles not re-read, 2 no ranking epilogue, 8 no record store
};

// rank_mode 1: bf16 x 3; 2: the stored vectors are bf16-exact (hi planes only); qlo: the batch's queries have a lo plane
// gq: queries per work item the grouping formed, 128 or 256
vi_status launch_rank_stream(const RankStreamArgs &a, uint32_t nc, uint32_t nitems, int rank_mode, bool qlo, uint32_t gq, hipStream_t st);

}  // namespace vi
