// rank_stream.hip — list ranking on the matrix cores, D <= 128: queries in LDS, vectors through registers.
//
// Replaces the workings of ivf_index.rs:205-262 (the per-list distance loop of `search_with_paths`) for a whole batch;
// what it produces — sub-block minima of m(q, v) = ||v||^2 - 2 q.v in pair records and the four smallest of a segment
// in group records — is what filter_search.hip's select turns into the exact top-k.
//
// Work item = one list segment x up to 128 queries that probe it (item_desc_kernel), one workgroup of 4 waves.
//
//   * B operand: the item's queries (-2 q, bf16 hi [+ lo]) are gathered ONCE into LDS by LDS-DMA with per-lane row
//     addresses, as [chunk][plane][half][query] x 16 B: a lane's fragment is one conflict-free ds_read_b128, and the
//     image is read-only until the item ends — the block loop has NO barrier.
//   * A operand: each wave streams its own 32-vector tiles (tile 4i + w of the segment) from the bf16 image straight
//     into registers with ordinary 16-byte loads, one tile ahead (8 KB per wave in flight, 64-96 KB per CU), and
//     multiplies a tile with every live 32-query tile of the item: 8 MFMAs (32x32x16) per query tile and plane
//     product; the accumulator starts at the norms (the first MFMA's C operand).
//   * the waves advance independently: a wave that waits for its tile leaves the matrix pipe to the others, nothing
//     waits for the slowest wave of a block, and a group with few queries costs MFMAs for its live query tiles only
//     (the block-synchronous kernel this replaces idled whole waves of a partially filled group).
//
// What was wrong with the block-synchronous kernel (filter_kernel, still used for the coarse table and the f32 MFMA):
// waves spent 47 % of their cycles in s_waitcnt (SQ_WAIT_ANY) — one 16 KB tile in flight per workgroup behind a
// barrier per block cannot cover an L2 / Infinity-Cache round trip with 512 MFMA cycles per wave and block.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#include "common.hpp"
#include "mfma_bf16.hpp"
#include "rank_stream.hpp"

namespace vi {
namespace {

constexpr int kWave = 64;
// phase clocks of wave 0 (VI_STREAM_PROF at run time) are compiled in only with -DVI_STREAM_PROF_BUILD: their counters
// cost 32 registers the kernel does not have
#ifdef VI_STREAM_PROF_BUILD
constexpr bool kStreamProf = true;
#else
constexpr bool kStreamProf = false;
#endif

// one 32-vector tile of the bf16 image in registers: chunk c, plane p at a[c * NA + p]
template <int NC, int NA>
struct TileRegs {
  uint4 a[NC * NA];
  float4 n[4];  // the 16 norms this lane's accumulator rows start from
};

template <int NC, int NA>
__device__ __forceinline__ void load_tile(TileRegs<NC, NA> &t, const uint4 *img, const float *xnorm, uint32_t blk, uint32_t half,
                                          int il, int h) {
  // image of a block: piece (chunk c, plane p, half h) = (c * 2 + p) * 2 + h, 64 columns x 16 B each
  const uint4 *ap = img + ((size_t)blk * (NC * 4) + h) * kWave + 32u * half + il;
#pragma unroll
  for (int c = 0; c < NC; ++c)
#pragma unroll
    for (int p = 0; p < NA; ++p) t.a[c * NA + p] = ap[(c * 4 + 2 * p) * kWave];
  const float *np = xnorm + (size_t)blk * kWave + 32u * half + 4 * h;
#pragma unroll
  for (int q4 = 0; q4 < 4; ++q4) t.n[q4] = *reinterpret_cast<const float4 *>(np + 8 * q4);
}

// LDS row of one query: R real 16-byte pieces in RP = pow2 >= R slots.  Piece pc of row q sits in slot swz(pc, q):
// 16 lanes that read the same piece of 16 consecutive rows then touch 16 different 16-byte bank groups
// (RP >= 16: the bank group is slot mod 16 -> XOR the row's low 4 bits; RP < 16: 16 / RP rows share 256 B ->
// XOR with (q / (16 / RP)) mod RP).  swz is its own inverse in pc.
template <int R>
struct StreamLayout {
  static constexpr int RP = R <= 2 ? 2 : R <= 4 ? 4 : R <= 8 ? 8 : R <= 16 ? 16 : 32;
  static_assert(R <= 32, "row too long");
  __device__ static __forceinline__ uint32_t swz(uint32_t pc, uint32_t q) {
    if constexpr (RP >= 16) return pc ^ (q & 15u);
    else return pc ^ ((q / (16u / RP)) & (uint32_t)(RP - 1));
  }
};

// two query images per workgroup when they fit beside a second workgroup on the CU
__host__ __device__ constexpr bool stream_double_buffered(int image_bytes) { return image_bytes <= 32 * 1024; }

// what an item's workgroup needs to know about it: requested an item ahead (ItemRaw: the loads' destination registers,
// untouched until the item's predecessor is in its last step) and decoded then
struct ItemRaw {
  uint4 d;          // {queries, first block, 32-vector tiles, first record tile} (item_cols_kernel)
  uint32_t qid;     // lane l < GQ / 4: query of row wave * GQ / 4 + l of the item's group (the rows this wave gathers), ~0: none
  uint32_t rec[2];  // group record (lane half 0) of columns 32 (wave + 4 r) + j, ~0: none
};
struct ItemRegs {
  uint32_t nqi, blk00, ntiles, rec0;  // wave-uniform
};

// "the tile has landed": an empty asm that reads every register of the tile makes the compiler place the wait for
// the tile's loads HERE, before the next tile's loads are issued.  Left to itself it waited with vmcnt(0) in front of
// the first MFMA of a step — after the prefetch of the next tile had been issued, i.e. for that one too.
template <int NC, int NA>
__device__ __forceinline__ void tile_landed(const TileRegs<NC, NA> &t) {
#pragma unroll
  for (int i = 0; i < NC * NA; ++i) asm volatile("" ::"v"(t.a[i].x), "v"(t.a[i].y), "v"(t.a[i].z), "v"(t.a[i].w));
#pragma unroll
  for (int q4 = 0; q4 < 4; ++q4) asm volatile("" ::"v"(t.n[q4].x), "v"(t.n[q4].y), "v"(t.n[q4].z), "v"(t.n[q4].w));
  __builtin_amdgcn_sched_barrier(0);
}

// s_waitcnt vmcnt(n) for a wave-uniform run-time n <= 8 (the instruction takes an immediate; a stricter wait is always safe)
__device__ __forceinline__ void wait_vmcnt(uint32_t n) {
  switch (n) {
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
}

// returning atomic increment whose result the CALLER waits for (s_waitcnt vmcnt) before using it
__device__ __forceinline__ uint32_t queue_pop_asm(uint32_t *counter) {
  uint32_t ret;
  const uint32_t one = 1u;
  asm volatile("global_atomic_add %0, %1, %2, off sc0" : "=v"(ret) : "v"(counter), "v"(one) : "memory");
  return ret;
}

// the same for the next item's description: placed right behind a tile's wait (its loads are older than the tile's)
__device__ __forceinline__ void item_landed(const ItemRaw &r) {
  asm volatile("" ::"v"(r.d.x), "v"(r.d.y), "v"(r.d.z), "v"(r.d.w), "v"(r.qid), "v"(r.rec[0]), "v"(r.rec[1]));
  __builtin_amdgcn_sched_barrier(0);
}

// RANK 1: bf16 x 3 (vector hi + lo planes, query hi + lo).  RANK 2: the stored vectors are bf16-exact (hi plane only);
// QLO says whether the batch's queries need their lo plane (false: every -2 q is bf16-exact too — 8-bit descriptors).
// NU = 32-query tiles per work item (4: groups of 128 queries; 8: groups of 256 — half the tile loads per query and
// twice the MFMAs behind every tile load, at 64 KB of LDS for hi-only query images).
//
// Persistent workgroups (two per CU) pull items from a counter.  An item's prologue used to be a chain of dependent
// loads — descriptor -> pairs -> query offsets -> gather — 6-7 us per item with nothing else to do, a third of the
// launch.  Now everything about an item is addressable from its index (item_desc_kernel, item_cols_kernel) and is
// loaded while the previous item is multiplied, as is the index of the item after it; only the gather of the queries
// into LDS (one level) is exposed between two items.
template <int NC, int RANK, bool QLO, int NU>
__global__ void __launch_bounds__(256, 2) rank_stream_kernel(RankStreamArgs a) {
  constexpr int NA = RANK == 1 ? 2 : 1;  // vector planes
  constexpr int NP = QLO ? 2 : 1;        // query planes
  constexpr int GQ = 32 * NU;            // queries per work item
  constexpr int NR = NU / 4;             // group-record columns per lane
  // LDS image of the item's queries: query-major, one row of RP 16-byte pieces per query (R = 2 NC NP of them real, in the
  // order of the global image: [plane][chunk][half]) — so that one LDS-DMA instruction reads 1 KB of whole cache lines
  // (a piece-major image took one 16-byte piece of 64 different rows per instruction).  A row's pieces are stored
  // XOR-swizzled (StreamLayout) so that 16 lanes reading the same piece of 16 consecutive rows hit 16 different bank
  // groups; the swizzle is applied by the gather on the SOURCE side (the DMA's LDS destination is lane order).
  using L = StreamLayout<2 * NC * NP>;
  constexpr int RP = L::RP;
  constexpr int RPI = 64 / RP;           // rows per LDS-DMA instruction
  constexpr int IPW = GQ * RP / 256;     // LDS-DMA instructions per wave and item
  constexpr int IMG = GQ * RP * 16;      // bytes of one image
  // two images when they fit: the next item's queries are gathered under the last steps of the current item
  constexpr bool DB = stream_double_buffered(IMG);
  static_assert(RANK == 2 || QLO, "bf16 x 3 needs the queries' lo plane");
  extern __shared__ __attribute__((aligned(16))) float s_mem[];
  float *s_T = s_mem + (DB ? 2 : 1) * (IMG / 4);             // [item parity][wave][query tile][lane]: the waves' minima of an item
  uint32_t *s_idx = reinterpret_cast<uint32_t *>(s_T + 2 * 4 * NU * kWave);  // item indices handed from wave 0 to the others ([parity])
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
  const int j = lane & 31, h = lane >> 5;
  const unsigned lds_q = (unsigned)(size_t)(lds_ptr_t)s_mem;  // LDS byte address of the query image(s)

  auto request_item = [&](uint32_t it, ItemRaw &r) {
    r.d = make_uint4(0u, 0u, 0u, 0u); r.qid = ~0u; r.rec[0] = ~0u; r.rec[1] = ~0u;
    if (it >= a.nitems) return;
    r.d = a.sdesc[it];
    if ((uint32_t)lane < (uint32_t)(GQ / 4)) r.qid = a.qcol[(size_t)it * GQ + (uint32_t)wave * (GQ / 4) + (uint32_t)lane];
#pragma unroll
    for (int k = 0; k < NR; ++k) r.rec[k] = a.grec[(size_t)it * GQ + 32u * ((uint32_t)wave + 4u * k) + (uint32_t)j];
  };
  auto decode_item = [&](const ItemRaw &r) {
    ItemRegs o;
    o.nqi = (uint32_t)__builtin_amdgcn_readfirstlane((int)r.d.x);
    o.blk00 = (uint32_t)__builtin_amdgcn_readfirstlane((int)r.d.y);
    o.ntiles = (uint32_t)__builtin_amdgcn_readfirstlane((int)r.d.z);  // 32-vector tiles of the segment; wave w owns tiles w, w + 4, ...
    o.rec0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)r.d.w);
    return o;
  };
  // an item's queries -> LDS image `buf`.  Wave w fills rows w GQ/4 .. + GQ/4 - 1, RPI rows per instruction: lane l =
  // (row, slot) fetches the piece that belongs in that slot; the row's query comes from the lane that holds it.
  auto gather = [&](uint32_t buf, uint32_t nq_item, uint32_t my_qid) {
    if (a.xmode & 32u) return;
    uint32_t lo = (uint32_t)lane;
    asm volatile("" : "+v"(lo));  // (keeps the per-instruction addresses from being computed once and held for the whole kernel)
    uint32_t qids[IPW];
#pragma unroll
    for (int i = 0; i < IPW; ++i) qids[i] = (uint32_t)__shfl((int)my_qid, (int)(RPI * i + lo / RP));
#pragma unroll
    for (int i = 0; i < IPW; ++i) {
      const uint32_t row0 = (uint32_t)wave * (GQ / 4) + (uint32_t)(RPI * i);
      if (row0 < nq_item) {  // wave-uniform
        const uint32_t pc = L::swz(lo % RP, row0 + lo / RP);
        if (qids[i] != ~0u && pc < 2u * NC * NP)
          glds16_at(a.qimg + (size_t)qids[i] * (NC * 4) + pc, lds_q + buf * (uint32_t)IMG + row0 * RP * 16u);
      }
    }
  };

  const unsigned long long rt_begin = a.prof ? __builtin_amdgcn_s_memrealtime() : 0ull;  // (100 MHz, chip-wide)
  // Eight queues, one per XCD (workgroup b runs on XCD b % 8): queue x holds items [x n / 8, (x + 1) n / 8).  Items are
  // numbered list by list, segment by segment, so the query groups of one list segment — which stream the same blocks —
  // are taken by workgroups of one XCD at about the same time and share its L2.  A queue that has run dry sends the
  // workgroup on to the next one.  Counter x at a.queue[32 x] (a cache line of its own).
  constexpr uint32_t kQueues = 8;
  auto queue_range = [&](uint32_t q, uint32_t &lo, uint32_t &hi) {
    lo = (uint32_t)(((uint64_t)a.nitems * q) / kQueues);
    hi = (uint32_t)(((uint64_t)a.nitems * (q + 1)) / kQueues);
  };
  uint32_t my_queue = blockIdx.x % kQueues;   // (thread 0's view: the queue it pops from)
  auto resolve = [&](uint32_t popped) {       // thread 0: popped value of my_queue -> item index, or the next queue's, ... or none
    for (uint32_t tries = 0; tries < kQueues; ++tries) {
      uint32_t lo, hi;
      queue_range(my_queue, lo, hi);
      if (popped < hi - lo) return lo + popped;
      my_queue = (my_queue + 1) % kQueues;
      if (tries + 1 < kQueues) popped = atomicAdd(a.queue + 32 * my_queue, 1u);
    }
    return a.nitems;  // every queue is empty
  };
  // ---- pipeline fill: two item indices, their descriptions, the first item's queries and first tile.  A workgroup holds
  //      its current item and the next; the one after is claimed during the current item's last step — items claimed
  //      early cannot be taken by a workgroup that runs dry, and the launch ends with its slowest workgroup ----
  const bool fixed = (a.xmode & 64u) != 0u;  // ablation: items dealt by stride instead of the counter
  if (threadIdx.x == 0) {
#pragma unroll
    for (uint32_t k = 0; k < 2; ++k) s_idx[k] = fixed ? blockIdx.x + k * gridDim.x : resolve(atomicAdd(a.queue + 32 * my_queue, 1u));
  }
  __syncthreads();
  uint32_t cur = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_idx[0]), nxt = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_idx[1]);
  ItemRaw rc, rn;
  request_item(cur, rc);
  request_item(nxt, rn);
  ItemRegs ic = decode_item(rc);
  TileRegs<NC, NA> ta, tb;
  if ((uint32_t)wave < ic.ntiles) load_tile<NC, NA>(ta, a.img, a.xnorm, ic.blk00 + ((uint32_t)wave >> 1), (uint32_t)wave & 1u, j, h);
  gather(0u, ic.nqi, rc.qid);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the LDS-DMA is invisible to the compiler's counters)
  __syncthreads();
  tile_landed<NC, NA>(ta);

  // diagnostic (-DVI_STREAM_PROF_BUILD, a.prof != null): s_memtime ticks wave 0 of this workgroup spent in [0] multiplying,
  // [1] waiting for the other waves at the item's end, [8..10, 2] between the barriers, [11, 12, 3] after them; [4] items
  unsigned long long pt[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  const bool prof = kStreamProf && a.prof != nullptr && wave == 0;
  unsigned long long tk = prof ? __builtin_amdgcn_s_memtime() : 0ull;
  const unsigned long long tk_begin = tk;
  auto lap = [&](int slot) {
    if (prof) {
      const unsigned long long now = __builtin_amdgcn_s_memtime();
      pt[slot] += now - tk;
      tk = now;
    }
  };
  uint32_t buf = 0;  // image of the current item (DB)
  uint32_t par = 0;  // parity of the current item (buffers of the item-end hand-over)
  uint32_t n_items = 0;
  // piece (plane p, chunk c, half h) of query 32 u + j: (row base | swizzle of the lane) XOR the piece (StreamLayout)
  const uint32_t lane_row = ((uint32_t)j * RP + L::swz(0u, (uint32_t)j)) * 16u;
  while (cur < a.nitems) {
    uint32_t after = ~0u;  // the item after the next (thread 0): claimed in this wave's last step, handed over at the item's end
    ItemRegs in{0u, 0u, 0u, 0u};
    // the next item's description (requested at the end of the previous item) is decoded, its queries are requested
    // and the item after it is claimed as late as this wave can afford: before its last step
    auto last_step = [&](TileRegs<NC, NA> &spare) {
      in = decode_item(rn);
      // this wave's first tile of the next item, into the tile registers the last step does not use: requested BEFORE the
      // step's record stores, so that waiting for it (and the queries) at the item's end does not wait for the stores
      if ((uint32_t)wave < in.ntiles) load_tile<NC, NA>(spare, a.img, a.xnorm, in.blk00 + ((uint32_t)wave >> 1), (uint32_t)wave & 1u, j, h);
      // (asm: the compiler turns atomicAdd under a one-lane branch into its wave-aggregated form, which waits for the
      // result — and every older load of the wave — on the spot; the result is needed at the item's end)
      if (threadIdx.x == 0) after = fixed ? nxt + gridDim.x : queue_pop_asm(a.queue + 32 * my_queue);
      if (DB) gather(buf ^ 1u, in.nqi, rn.qid);
    };
    const uint32_t nqi = ic.nqi, nu = (ic.nqi + 31u) >> 5, ntiles = ic.ntiles, blk00 = ic.blk00;
    const char *img_at = reinterpret_cast<const char *>(s_mem) + (DB ? buf * (uint32_t)IMG : 0u);
    auto frag = [&](int c, int p, int u) {
      const uint32_t pc = (uint32_t)(p * 2 * NC + 2 * c) + (uint32_t)h;
      return __builtin_bit_cast(bf16x8, *reinterpret_cast<const float4 *>(img_at + ((lane_row ^ (pc << 4)) + (uint32_t)(32 * u * RP * 16))));
    };

    float T[NU];       // per query tile: the smallest sub-block minimum among this wave's tiles
    float4 pend[NU];   // per query tile: the pair record being filled (one component per tile of this wave)
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      T[u] = INFINITY;
      pend[u] = make_float4(INFINITY, INFINITY, INFINITY, INFINITY);
    }
    // pair records: [record rt = (i >> 2) * 4 + wave of the segment][lane half][query of the group] x 16 B; component i & 3
    // = tile 4 i + wave (i = this wave's step): a wave stores whole 16-byte records, 512 contiguous bytes per lane half
    const uint32_t bi = ic.rec0 * (2u * GQ) + (uint32_t)GQ * (uint32_t)h + (uint32_t)j;

    // Query fragments run kPF chunks ahead of the MFMAs that consume them, in a ring of registers, across the query
    // tiles of a step (position s = u * NC + c): left to itself the compiler reads each fragment right before its MFMA
    // and the wave stalls on the LDS latency every time.  Positions past the last live query tile read a clamped
    // (valid, unused) address.
    constexpr int kPF = 4;
    auto step = [&](const TileRegs<NC, NA> &t, uint32_t i, bool last) {
      f32x16 nrm;
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        nrm[4 * q4 + 0] = t.n[q4].x; nrm[4 * q4 + 1] = t.n[q4].y; nrm[4 * q4 + 2] = t.n[q4].z; nrm[4 * q4 + 3] = t.n[q4].w;
      }
      bf16x8 ring[kPF][NP];
      auto fetch = [&](int s) {
        const int u = (s / NC) < NU - 1 ? (s / NC) : NU - 1, c = s % NC;
#pragma unroll
        for (int p = 0; p < NP; ++p) ring[s % kPF][p] = frag(c, p, u);
      };
#pragma unroll
      for (int s0 = 0; s0 < kPF; ++s0) fetch(s0);
      const uint32_t ci = i & 3u;
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        if ((uint32_t)u < nu) {  // wave-uniform
          f32x16 acc = nrm;
#pragma unroll
          for (int c = 0; c < NC; ++c) {
            const int sp = u * NC + c;
            const bf16x8 ah = __builtin_bit_cast(bf16x8, t.a[c * NA]);
            const bf16x8 bh = ring[sp % kPF][0];
            __builtin_amdgcn_sched_barrier(0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
            if constexpr (QLO) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, ring[sp % kPF][NP - 1], acc, 0, 0, 0);
            if constexpr (RANK == 1) {
              const bf16x8 al = __builtin_bit_cast(bf16x8, t.a[c * NA + 1]);
              acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            fetch(sp + kPF);  // into the slot this chunk has just been issued from
          }
          if (!(a.xmode & 2u)) {
            const float m = tile_min(acc);
            T[u] = min3_raw(T[u], m, m);
            pend[u].x = ci == 0u ? m : pend[u].x;
            pend[u].y = ci == 1u ? m : pend[u].y;
            pend[u].z = ci == 2u ? m : pend[u].z;
            pend[u].w = ci == 3u ? m : pend[u].w;
          }
        }
      }
      if (ci == 3u || last) {  // the record is complete (wave-uniform)
        const uint32_t rt = (i >> 2) * 4u + (uint32_t)wave;
#pragma unroll
        for (int u = 0; u < NU; ++u) {
          if ((uint32_t)u < nu) {
            if (32u * u + (uint32_t)j < nqi && !(a.xmode & 8u)) a.brec[(size_t)bi + 32u * u + (2u * GQ) * rt] = pend[u];
            pend[u] = make_float4(INFINITY, INFINITY, INFINITY, INFINITY);
          }
        }
      }
    };

    // ---- the wave's tiles, one ahead: registers ta / tb alternate.  With two images the next item's queries are
    //      requested right before the wave's last step: their latency runs under that step and the wait for the
    //      other waves ----
    const bool restage = !(a.xmode & 1u);
    bool prepared = false, next_in_tb = false;
    for (uint32_t i = 0;; i += 2) {
      const uint32_t t0 = 4u * i + (uint32_t)wave;
      if (t0 >= ntiles || (a.xmode & 16u)) break;
      const uint32_t t1 = t0 + 4u, t2 = t0 + 8u;
      tile_landed<NC, NA>(ta);
      item_landed(rn);  // (older than the tile: free)
      if (t1 < ntiles && restage) load_tile<NC, NA>(tb, a.img, a.xnorm, blk00 + (t1 >> 1), t1 & 1u, j, h);
      if (t1 >= ntiles) { last_step(tb); prepared = true; next_in_tb = true; }
      step(ta, i, t1 >= ntiles);
      if (t1 >= ntiles) break;
      tile_landed<NC, NA>(tb);
      item_landed(rn);
      if (t2 < ntiles && restage) load_tile<NC, NA>(ta, a.img, a.xnorm, blk00 + (t2 >> 1), t2 & 1u, j, h);
      if (t2 >= ntiles) { last_step(ta); prepared = true; }
      step(tb, i + 1u, t2 >= ntiles);  // (ablation xmode 1: whatever the registers hold)
    }
    if (!prepared) last_step(ta);  // (a wave without tiles in this item)
    // record stores this wave issued after those requests (its last step's, one per live query tile)
    const uint32_t young_stores = (ntiles > (uint32_t)wave && !(a.xmode & (8u | 16u))) ? nu : 0u;

    // ---- item end.  Group records: their four values are the minima of the four waves' tile classes (tiles = w mod
    //      4 of the segment), sorted: four distinct sub-blocks' minima, the smallest of them the segment's minimum —
    //      all the select relies on (a bound from K listed values holds for any K distinct sub-blocks; pair records
    //      list the rest).  The segment's true four smallest would cost four registers per query tile and wave.
    // One barrier per item with two images: a wave requests the queries of item k + 2 (into item k's image) in its last
    // step of item k + 1, behind the barrier that ended item k for everyone; the minima and the handed-over index go
    // through buffers of the item's parity, so a fast wave's next round cannot overtake a slow wave's reads.
    lap(0);
    if (!DB) __syncthreads();  // (one image: every wave is done with it before the next item's queries overwrite it)
    lap(1);
    float *s_Tk = s_T + par * (4 * NU * kWave);
#pragma unroll
    for (int u = 0; u < NU; ++u)
      if ((uint32_t)u < nu) s_Tk[(wave * NU + u) * kWave + lane] = T[u];
    lap(8);
    if (!DB) gather(0u, in.nqi, rn.qid);
    lap(9);
    // the next item's queries and first tile have landed; with two images they are older than the last step's stores
    if (DB) wait_vmcnt(young_stores);
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (threadIdx.x == 0) s_idx[par] = fixed ? after : resolve(after);  // (behind the wait: queue_pop_asm's result has arrived)
    lap(10);
    __syncthreads();
    lap(2);
    if (next_in_tb) ta = tb;
    // (nothing of this wave is in flight here: the compiler's wait for rc.rec — it cannot know that — belongs HERE and
    // not between the two record stores below, where it would wait for the first store to complete)
    asm volatile("" ::"v"(rc.rec[0]), "v"(rc.rec[1]));
    tile_landed<NC, NA>(ta);  // (the next item's first tile too: its first step must not wait behind the stores below)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const uint32_t u = (uint32_t)wave + 4u * r;
      if (u < nu) {
        float v0 = s_Tk[(0 * NU + u) * kWave + lane], v1 = s_Tk[(1 * NU + u) * kWave + lane];
        float v2 = s_Tk[(2 * NU + u) * kWave + lane], v3 = s_Tk[(3 * NU + u) * kWave + lane];
        auto cx = [](float &x, float &y) { const float lo = fminf(x, y), hi = fmaxf(x, y); x = lo; y = hi; };
        cx(v0, v1); cx(v2, v3); cx(v0, v2); cx(v1, v3); cx(v1, v2);  // sort four
        if (rc.rec[r] != ~0u) a.gval[rc.rec[r] + (uint32_t)h] = make_float4(v0, v1, v2, v3);
      }
    }
    lap(11);
    cur = nxt;
    nxt = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_idx[par]);
    par ^= 1u;
    ic = in;
    rc = rn;
    buf ^= 1u;
    lap(12);
    request_item(nxt, rn);
    lap(3);
    pt[4] += 1;
    ++n_items;
  }
  if (a.prof && threadIdx.x == 0) {  // per workgroup: start, end (100 MHz ticks), items
    a.prof[32 + 4 * blockIdx.x + 0] = rt_begin;
    a.prof[32 + 4 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
    a.prof[32 + 4 * blockIdx.x + 2] = n_items;
  }
  if (prof && lane == 0) {
    pt[7] = __builtin_amdgcn_s_memtime() - tk_begin;
#pragma unroll
    for (int k = 0; k < 13; ++k) atomicAdd(a.prof + k, pt[k]);
    atomicMax(a.prof + 13, pt[7]);                 // longest loop of a workgroup
    atomicAdd(a.prof + 14, pt[4] ? 1ull : 0ull);   // workgroups that had an item
    atomicMax(a.prof + 15, pt[4]);                 // most items of a workgroup
    a.prof[32 + 4 * blockIdx.x + 3] = pt[0];
    atomicMin(a.prof + 16, tk_begin);              // first / last workgroup to enter its loop, last to leave (absolute ticks)
    atomicMax(a.prof + 17, tk_begin);
    atomicMax(a.prof + 18, tk_begin + pt[7]);
  }
}

template <int NC, int RANK, bool QLO, int NU>
vi_status launch_one(const RankStreamArgs &a, uint32_t nitems, hipStream_t st) {
  // LDS: the query image, the waves' minima of an item (1 KB per query tile), the item indices
  const size_t img = (size_t)StreamLayout<2 * NC * (QLO ? 2 : 1)>::RP * (32 * NU) * 16;
  const size_t lds = img * (stream_double_buffered((int)img) ? 2 : 1) + 2 * (size_t)NU * 1024 + 16;
  // per device (a process may hold indexes on several GPUs: the attribute and the CU count belong to the device that launches)
  constexpr int kMaxDev = 64;
  static uint32_t cus_of[kMaxDev];  // 0 = not yet prepared on that device
  int dev = 0;
  VI_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= kMaxDev) return fail(VI_ERR_DEVICE, "device ordinal %d out of range", dev);
  uint32_t cus = __atomic_load_n(&cus_of[dev], __ATOMIC_ACQUIRE);
  if (cus == 0) {
    if (hipFuncSetAttribute((const void *)rank_stream_kernel<NC, RANK, QLO, NU>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return fail(VI_ERR_DEVICE, "cannot reserve %zu bytes of LDS for the rank kernel", lds);
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cus = (uint32_t)n;
    __atomic_store_n(&cus_of[dev], cus, __ATOMIC_RELEASE);
  }
  uint32_t per_cu = lds > 80 * 1024 ? 1u : 2u;  // persistent workgroups: as many as fit on the chip at once
  if (const char *e = getenv("VI_STREAM_WGS_PER_CU")) per_cu = (uint32_t)std::max(1, atoi(e));  // (experiment)
  if (a.prof) {
    int nb = -1;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)rank_stream_kernel<NC, RANK, QLO, NU>, 256, lds);
    fprintf(stderr, "rank_stream<%d,%d,%d,%d>: lds %zu B, occupancy %d workgroups per CU, grid %u\n", NC, RANK, (int)QLO, NU, lds, nb,
            std::min(nitems, per_cu * cus));
  }
  hipLaunchKernelGGL((rank_stream_kernel<NC, RANK, QLO, NU>), dim3(std::min(nitems, per_cu * cus)), dim3(256), lds, st, a);
  VI_HIP(hipGetLastError());
  return VI_OK;
}

template <int NC>
vi_status launch_nc(const RankStreamArgs &a, uint32_t nitems, int rank_mode, bool qlo, uint32_t gq, hipStream_t st) {
  // (bf16 x 3 — RANK 1 — compiles but needs two tiles of hi + lo planes in registers: it spills; the pipeline keeps the
  // block-synchronous kernel for it and does not instantiate it here)
  if (rank_mode != 2) return fail(VI_ERR_OTHER, "the streaming rank kernel is built for bf16-exact lists only");
  if (gq == 256) {  // groups of 256 are formed for batches of bf16-exact queries only (filter_search.hip)
    if (!qlo) return launch_one<NC, 2, false, 8>(a, nitems, st);
    return launch_one<NC, 2, true, 8>(a, nitems, st);
  }
  if (!qlo) return launch_one<NC, 2, false, 4>(a, nitems, st);
  return launch_one<NC, 2, true, 4>(a, nitems, st);
}

}  // namespace

vi_status launch_rank_stream(const RankStreamArgs &a, uint32_t nc, uint32_t nitems, int rank_mode, bool qlo, uint32_t gq, hipStream_t st) {
  if (nitems == 0) return VI_OK;
  switch (nc) {
    case 1: return launch_nc<1>(a, nitems, rank_mode, qlo, gq, st);
    case 2: return launch_nc<2>(a, nitems, rank_mode, qlo, gq, st);
    case 3: return launch_nc<3>(a, nitems, rank_mode, qlo, gq, st);
    case 4: return launch_nc<4>(a, nitems, rank_mode, qlo, gq, st);
    case 5: return launch_nc<5>(a, nitems, rank_mode, qlo, gq, st);
    case 6: return launch_nc<6>(a, nitems, rank_mode, qlo, gq, st);
    case 7: return launch_nc<7>(a, nitems, rank_mode, qlo, gq, st);
    case 8: return launch_nc<8>(a, nitems, rank_mode, qlo, gq, st);
    default: return fail(VI_ERR_OTHER, "unsupported dimension for the streaming rank kernel");
  }
}

}  // namespace vi
