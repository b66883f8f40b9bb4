// device_index.hpp — HBM-resident IVF index and the search pipeline entry points.
//
// HBM layout ("lane-interleaved blocks").  Every vector set that gets scanned — the coarse
// centroid table and each inverted list — is stored in blocks of 64 vectors:
//
//     block[b] : float4 [dq][64]          dq = 4*ceil(dim/16) quads, zero padded
//     element (qd, lane).{x,y,z,w} = dims 4*qd .. 4*qd+3 of vector (64*b + lane)
//
// so that a wave64 reading quad qd of a block issues ONE global_load_dwordx4 covering a
// contiguous, 1 KiB-aligned span: lane = vector, registers = consecutive dimensions.  That
// makes the reference's strictly sequential f32 sum over d (src/utils.rs:28-30) a per-lane
// register chain with perfectly coalesced loads, and lets one loaded quad be reused by a
// whole group of queries from registers.  Lists are padded to whole blocks (pad lanes carry
// zeros and are masked by the list length).  External ids live in a parallel u64 array
// indexed by slot = 64*block + lane.  The AoS record layout of the shard files
// (24 B meta + 4·D B + pad, shards.rs:106-114) is kept only on disk.
#pragma once
#include <cstdint>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "common.hpp"
#include "shards.hpp"

namespace vi {

constexpr uint32_t kNoPos = 0xFFFFFFFFu;   // empty slot marker in a sorted run
constexpr uint32_t kMaxSelect = 64;        // wave-resident top-k width of the fast path

struct BlockSet {               // a set of lists stored as lane-interleaved blocks
  DevBuf<float> blocks;         // [nblocks][dq][64][4]
  uint32_t dq = 0;
  uint64_t nblocks = 0;
};

struct SearchWorkspace {
  DevBuf<float> q;              // queries on device (nq x dim) when the caller passed host memory
  DevBuf<float> crun_dist;      // coarse partial runs [nq][S][P]
  DevBuf<uint32_t> crun_pos;
  DevBuf<uint32_t> probes;      // [nq][P] list ids in probe-rank order
  DevBuf<uint32_t> gorder;      // [nq][P] candidate-order rank of each probe (shard visiting order)
  DevBuf<uint32_t> probe_flag;  // validation result of caller-supplied probe lists
  DevBuf<uint32_t> cnt;         // [nlists] (#queries probing list) ; cursor = second half
  DevBuf<uint32_t> list_tot;    // [nlists] queries probing each list
  DevBuf<uint32_t> seg_start;   // [nlists+1]
  DevBuf<uint32_t> item_start;  // [nlists+1]
  DevBuf<uint32_t> segrun_start;  // [nlists+1]
  DevBuf<float> seg_run_dist;   // segment runs of long lists, merged by seg_merge_kernel
  DevBuf<uint32_t> seg_run_pos;
  DevBuf<uint32_t> pairs;       // [nq*P] slot ids grouped by list
  DevBuf<float> run_dist;       // [nq*P][K]
  DevBuf<uint32_t> run_pos;
  DevBuf<float> D;              // outputs when the caller wants host results
  DevBuf<int64_t> I;
  DevBuf<uint64_t> tie;
  DevBuf<uint64_t> slots;       // [nq][k] global slot of each result (include_vectors gather)
  DevBuf<uint32_t> counts;      // [nq]
  DevBuf<uint64_t> stats;       // device-side counters
  DevBuf<float> V;
  // MFMA filter path (filter_search.hip)
  DevBuf<uint32_t> c_seg, c_item, c_pairs;  // the coarse table grouped as one list ...
  uint64_t c_nq = 0;                        // ... for this batch size
  DevBuf<uint32_t> pair_rel, qtot, qoff;    // group-record offsets: per (query, probe), per query, scan over queries
  DevBuf<uint32_t> pair_rank;               // a pair's place among the pairs of its (list, sub-bin): from the direct coarse select
  bool pair_rank_valid = false;             // ... filled by this search
  DevBuf<uint32_t> item_list;               // list of each rank work item
  DevBuf<uint32_t> items;                   // ... or its whole descriptor (8 words), item_desc_kernel
  DevBuf<uint64_t> prof;                    // diagnostic counters (VI_STREAM_PROF)
  DevBuf<uint32_t> item_qcol, item_grec, item_sdesc;    // streaming rank kernel: per (item, column) the query / its group record (item_cols_kernel)
  DevBuf<uint32_t> tile_start, pair_pos;    // pair records: first record tile of each list; position of a (query, probe) pair in its list
  DevBuf<float> gval;                       // group records: 4 smallest sub-block minima per (query, probe, segment, lane half)
  DevBuf<uint32_t> gpos;                    // ... and where each record belongs (probe rank | segment | lane half)
  DevBuf<uint32_t> qimg;                    // wide vectors: query-major bf16 hi/lo image of the batch
  DevBuf<float> brec;                       // pair records: the 4 sub-block minima of two blocks per (record tile, lane half, query of the group)
  struct GqHint { uint64_t nq; uint32_t P, gq; };
  std::vector<GqHint> gq_hint;              // queries per rank work item measured to suit a batch shape (filter_search.hip)
  uint64_t *hstats_pinned = nullptr;        // page-locked landing buffer of the grouping's counts (16 words)
  bool stats_zeroed = false;                // stats[13], [14] start at zero (filter_search.hip)
  bool queries_hi_only = false;             // the previous batch's -2 q were all bf16-exact (no lo plane)
  DevBuf<uint64_t> sort_keys, order_keys, total;
  DevBuf<uint32_t> gprobe, off_by_g, off_by_rank;
};

struct DeviceIndex {
  int device = 0;
  int order = VI_ORDER_SCALAR;  // summation order of every distance this index computes
  uint32_t dim = 0, dq = 0;
  uint64_t nlists = 0;          // k' (non-empty centroids of the index)
  uint64_t nvec_resident = 0;   // vectors whose lists are resident on this GPU
  uint64_t nshards = 0;
  BlockSet centroids;           // the coarse table as one list of k' vectors
  BlockSet lists;               // all resident inverted lists back to back
  DevBuf<uint32_t> list_first_block;  // [nlists]
  DevBuf<uint32_t> list_len;          // [nlists]  (0 => not resident here / empty)
  DevBuf<uint32_t> list_shard;        // [nlists]
  DevBuf<uint64_t> ext_ids;           // [lists.nblocks*64]
  uint32_t stripe_rank = 0, stripe_world = 1;  // multi-GPU: block b of a list lives on rank b % world
  DevBuf<float> xnorm;                // [lists.nblocks*64] squared norm per slot (3e38 on pad slots)
  DevBuf<float> xnorm_img, cent_xnorm_img;  // the same in the column order of the bf16 images (filter_search.hip: image_column)
  DevBuf<uint32_t> lists_bf16, cent_bf16;  // bf16 hi/lo images of the blocks for the MFMA ranking (filter_search.hip)
  DevBuf<uint32_t> lists_u8_nat;           // 8-bit descriptors (integers 0..255): one byte per dimension, same order (exact re-evaluation)
  DevBuf<uint32_t> lists_hi_nat;           // bf16-exact lists: their hi plane in the blocks' own vector order (exact re-evaluation)
  bool lists_lo_zero = false, cent_lo_zero = false;  // every stored value is bf16-exact (lo planes all zero)
  // sampled means over the stored vectors: ||v - centroid of its list||^2 and ||v||^2 (what the ranking arithmetic of
  // real-valued lists is chosen by: filter_search.hip, rank_approx_mode)
  float mean_spread = 0.0f, mean_norm2 = 0.0f;
  float rho2_max = 0.0f;  // max |x - hi(x)|^2 over the stored vectors (x = v - mu when centred): the hi-plane ranking's margin
  // Real-valued lists far from the origin (SIFT-like values with noise, embeddings with a common offset): the bf16 images
  // are taken about the mean mu of the stored vectors — ||q - v|| does not change, the norms the rank margins scale with
  // shrink to the spread of the data.  Only the ranking sees mu; the exact evaluation reads the stored f32 values.
  bool centered = false;
  DevBuf<float> centre;  // mu: dq * 4 floats (zero beyond dim)
  float mean_norm2_c = 0.0f, xmax2_c = 0.0f, cent_xmax2_c = 0.0f;  // ... and the norms about mu
  float xmax2 = 0.0f;                 // max squared norm of a stored vector (MFMA filter margin)
  DevBuf<float> cent_xnorm;           // same for the coarse table
  DevBuf<float> cent_rows;            // the coarse table row-major (coarse select: single-row exact re-evaluation)
  float cent_xmax2 = 0.0f;
  DevBuf<uint32_t> c_first, c_len;    // the coarse table described as one list
  hipStream_t stream = nullptr;       // uploads and index construction
  // Searches: the reference's search is &self and runs concurrently from several OS threads
  // (tests/ivf_index_tests.rs:768-807).  A search holds one SearchContext — its own stream, events, workspace — for
  // the duration of the call; up to kSearchContexts calls run at once on one handle, further callers wait for a free one.
  struct SearchContext {
    hipStream_t stream = nullptr;
    hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    SearchWorkspace ws;
    vi_search_stats stats{};
    ~SearchContext();
  };
  static constexpr int kSearchContexts = 4;
  mutable std::mutex mu;              // guards the context pool and last_stats
  mutable std::condition_variable cv;
  mutable std::vector<std::unique_ptr<SearchContext>> contexts;  // created on demand
  mutable std::vector<SearchContext *> free_contexts;
  mutable vi_search_stats last_stats{};  // of the most recent search that finished on this handle
  SearchContext &cur() const;         // the context the calling thread holds (device_index_search acquired it)
  int timing = 0;  // 0 off, 1 HIP events at every phase boundary, 2 around the list-rank kernel only

  ~DeviceIndex();
};

// Upload: centroid table + this rank's part of the lists, repacked on the GPU.  placement 0: a stripe of every list
// (block b on rank b % world); 1: the whole lists of the shard files dealt to this rank (greedy by bytes).
vi_status device_index_load(const IndexMeta &meta, const std::string &shards_dir, int device, int rank,
                            int world, int placement, DeviceIndex *out);
// owner rank of every shard under placement 1: shards by descending bytes, each to the least loaded rank so far
std::vector<uint32_t> shard_owners(const std::vector<uint64_t> &shard_bytes, uint32_t world);

struct SearchIO {
  const float *queries = nullptr;  // host or device (queries_on_device)
  bool on_device = false;          // queries and D/I/tie are device pointers
  uint64_t nq = 0, k = 0, n_probe = 0;
  float *D = nullptr;
  int64_t *I = nullptr;
  uint64_t *tie = nullptr;
  float *V = nullptr;              // host only
  uint64_t *counts = nullptr;      // host only
  // multi-GPU: the coarse step of a query slice can run on another rank (device pointers, [nq][n_probe_eff])
  const uint32_t *probes_in = nullptr, *order_in = nullptr;  // skip the coarse step, use these probe lists
  uint32_t *probes_out = nullptr, *order_out = nullptr;      // coarse step only: probe lists + candidate-order ranks
};
vi_status device_index_search(const DeviceIndex &ix, const SearchIO &io);
// squared norms of the stored vectors (filter_search.hip); called at the end of every index upload
vi_status compute_slot_norms(DeviceIndex *ix);

// Build a device index from device-resident arrays (used by the k-means path, where the
// "index" is the two-level centroid hierarchy of assign_points_hierarchical, kmeans.rs:474-581,
// and by the GPU list build).  table: ntable x dim row-major (device) = coarse table;
// rows: row-major device matrix the lists draw from; list_off[nlists+1] (host) delimits
// member_rows (host, row index per list member, list order = scan order); ids (device,
// optional) = external id per row (null => row index); list_shard (host, optional).
vi_status device_index_from_rows(int device, int order, uint32_t dim, const float *table_dev, uint64_t ntable,
                                 const float *rows_dev, const std::vector<uint64_t> &list_off,
                                 const std::vector<uint32_t> &member_rows, const uint64_t *ids_dev,
                                 const std::vector<uint32_t> *list_shard, DeviceIndex *out);

// ---- GPU list build (list_build.hip) ----
// ids 0..n-1 grouped by label, ascending id inside a label (stable radix sort): order (device, n u32), seg (device,
// k+1 u32 offsets into order) and the offsets on the host if off_host is given.  Complete on return.  scratch_keep
// (optional): the sort's scratch (3 n + 256 n / 4096 words) in a buffer of the caller's, kept from call to call.
vi_status group_ids_by_label_device(const uint32_t *labels_dev, uint64_t n, uint64_t k, DevBuf<uint32_t> &order,
                                    DevBuf<uint32_t> &seg, std::vector<uint64_t> *off_host, hipStream_t st,
                                    DevBuf<uint32_t> *scratch_keep = nullptr);
// resident index straight from device data: list l = rows order[src_off[l] .. + len[l]) of X_dev
vi_status device_index_from_order(int device, uint32_t dim, const float *table_host, uint64_t nlists, const float *X_dev,
                                  const uint32_t *order_dev, const std::vector<uint64_t> &src_off,
                                  const std::vector<uint32_t> &len, const std::vector<uint32_t> &list_shard,
                                  const uint64_t *ids_dev, DeviceIndex *out);
struct ShardExportWs {  // staging of shard_export_device, reused from shard to shard
  DevBuf<uint8_t> image;
  DevBuf<uint64_t> d_src, d_dst;
  DevBuf<uint32_t> d_len, d_first;
  uint8_t *host = nullptr;  // pinned
  uint64_t host_cap = 0;
  ~ShardExportWs();
};
// shard_<id>.bin from device-resident points, byte-identical to shard_save_to (shards.cpp)
vi_status shard_export_device(const std::string &shards_dir, uint64_t shard_id, uint32_t dim, const std::vector<uint64_t> &cids,
                              const float *cvecs_host, const std::vector<uint64_t> &src_off, const std::vector<uint32_t> &len,
                              const float *X_dev, const uint32_t *order_dev, const uint64_t *ext_dev, const uint64_t *ts_dev,
                              uint64_t now, ShardExportWs &ws, hipStream_t st);

vi_status merge_partials_packed_device(int device, uint64_t nq, uint64_t k, uint32_t parts, const void *packed,
                                       float *D_out, int64_t *I_out);
vi_status merge_partials_device(int device, uint64_t nq, uint64_t k, uint32_t parts, const float *D_parts,
                                const int64_t *I_parts, const uint64_t *tie_parts, float *D_out,
                                int64_t *I_out);

vi_status l2sq_pairs_device(const float *a, const float *b, uint64_t n, uint32_t d, int order, float *out);

}  // namespace vi
