// shards.hpp — shard_<id>.bin on-disk format (byte-identical to the reference writer,
// src/shards.rs:22-51,68-177) and index/index.bin (src/ivf_index.rs:274-316).
//
// The reference re-opens and re-reads shard files with io_uring on every query
// (shards.rs:188-349).  Here files are only touched at build/load time: lists are
// repacked into the lane-interleaved HBM layout (device_index.hpp) and stay resident.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "common.hpp"

namespace vi {

// Layout constants of the repr(C) structs (shards.rs:22-51).
constexpr uint64_t kShardHeaderBytes = 40;  // the reference's "// 48 bytes" comment is wrong
constexpr uint64_t kIndexEntryBytes = 32;
constexpr uint64_t kVectorMetaBytes = 24;

inline uint64_t pad8(uint64_t vec_bytes) { return (8 - (vec_bytes % 8)) % 8; }
// bytes of one stored record: VectorMeta + D*f32 + pad (shards.rs:106-114)
inline uint64_t record_stride(uint32_t dim) { return kVectorMetaBytes + 4ull * dim + pad8(4ull * dim); }

struct ShardListView {       // one inverted list inside a mapped shard file
  uint64_t centroid_id = 0;
  uint32_t num_vectors = 0;
  const uint8_t *centroid = nullptr;  // dim f32
  const uint8_t *records = nullptr;   // num_vectors records of record_stride(dim) bytes
};

// Read-only memory map of one shard file with a validated index.
class ShardFile {
 public:
  ShardFile() = default;
  ~ShardFile();
  ShardFile(const ShardFile &) = delete;
  ShardFile &operator=(const ShardFile &) = delete;

  // Errors follow get_centroid_vectors_from (shards.rs:193-231): open failure => Other,
  // short/invalid header or shard-id mismatch => InvalidData.
  vi_status open(const std::string &shards_dir, uint64_t shard_id);
  uint32_t dim() const { return dim_; }
  uint32_t num_lists() const { return (uint32_t)lists_.size(); }
  const ShardListView &list(uint32_t i) const { return lists_[i]; }
  // linear find, as the reference does (shards.rs:257-265); nullptr => NotFound
  const ShardListView *find(uint64_t centroid_id) const;

 private:
  void *map_ = nullptr;
  size_t len_ = 0;
  uint32_t dim_ = 0;
  std::vector<ShardListView> lists_;
};

vi_status shard_save_to(const std::string &shards_dir, uint64_t shard_id, uint32_t dim,
                        uint32_t num_lists, const uint64_t *centroid_ids, const float *centroid_vecs,
                        const uint64_t *list_off, const uint64_t *ids, const uint64_t *ext_ids,
                        const uint64_t *timestamps, const float *vecs);

// index/index.bin: IvfIndex{centroids, centroids_to_shard, dimension} through bincode 2
// `config::standard()` + ndarray serde.  PARITY UNPINNED (no cargo here to confirm bytes).
struct IndexMeta {
  uint32_t dimension = 0;
  std::vector<float> centroids;      // k' x dimension
  std::vector<uint64_t> c2s;         // k'
  uint64_t k() const { return c2s.size(); }
};
vi_status index_meta_save(const IndexMeta &m, const std::string &index_dir);
vi_status index_meta_load(const std::string &index_dir, IndexMeta *out);

// batched vector file of read_vectors_from_file (src/utils.rs:82-107): concatenated
// bincode Vec<(u64, Vec<f32>, u64)> batches.  Decoding stops silently at the first
// undecodable batch (utils.rs:102), as the reference does.
struct VectorFileRecord { uint64_t id; std::vector<float> values; uint64_t meta; };
vi_status read_vectors_from_file(const std::string &path, std::vector<VectorFileRecord> *out);

vi_status make_dirs(const std::string &path);

}  // namespace vi
