// shards.cpp — see shards.hpp.
#include "shards.hpp"

#include <errno.h>
#include <fcntl.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstdio>
#include <cstring>

namespace vi {

namespace {
template <typename T>
T load_le(const uint8_t *p) {
  T v;
  std::memcpy(&v, p, sizeof(T));
  return v;
}
std::string shard_path(const std::string &dir, uint64_t id) {
  return dir + "/shard_" + std::to_string(id) + ".bin";
}
}  // namespace

vi_status make_dirs(const std::string &path) {
  if (path.empty()) return fail(VI_ERR_IO, "empty directory path");
  std::string cur;
  size_t i = 0;
  while (i <= path.size()) {
    if (i == path.size() || path[i] == '/') {
      if (!cur.empty() && mkdir(cur.c_str(), 0777) != 0 && errno != EEXIST)
        return fail(VI_ERR_IO, "create_dir_all(%s): %s", cur.c_str(), strerror(errno));
    }
    if (i < path.size()) cur.push_back(path[i]);
    ++i;
  }
  return VI_OK;
}

ShardFile::~ShardFile() {
  if (map_ && len_) munmap(map_, len_);
}

vi_status ShardFile::open(const std::string &shards_dir, uint64_t shard_id) {
  const std::string path = shard_path(shards_dir, shard_id);
  int fd = ::open(path.c_str(), O_RDONLY);
  if (fd < 0)
    return fail(VI_ERR_OTHER, "Failed to open shard_%llu.bin file: %s", (unsigned long long)shard_id,
                strerror(errno));
  struct stat st;
  if (fstat(fd, &st) != 0) {
    ::close(fd);
    return fail(VI_ERR_OTHER, "Failed to read header of shard_%llu.bin file", (unsigned long long)shard_id);
  }
  len_ = (size_t)st.st_size;
  if (len_ < kShardHeaderBytes) {
    ::close(fd);
    len_ = 0;
    return fail(VI_ERR_INVALID_DATA, "Invalid shard header, reading shard_%llu.bin.", (unsigned long long)shard_id);
  }
  map_ = mmap(nullptr, len_, PROT_READ, MAP_PRIVATE, fd, 0);
  ::close(fd);
  if (map_ == MAP_FAILED) {
    map_ = nullptr;
    len_ = 0;
    return fail(VI_ERR_OTHER, "mmap shard_%llu.bin: %s", (unsigned long long)shard_id, strerror(errno));
  }
  const uint8_t *base = (const uint8_t *)map_;
  const uint64_t file_shard = load_le<uint64_t>(base + 0);
  if (file_shard != shard_id)  // shards.rs:223-231
    return fail(VI_ERR_INVALID_DATA, "Shard ID mismatch: expected %llu, found %llu in file",
                (unsigned long long)shard_id, (unsigned long long)file_shard);
  dim_ = load_le<uint32_t>(base + 16);
  const uint32_t nc = load_le<uint32_t>(base + 20);
  const uint64_t index_off = load_le<uint64_t>(base + 24);
  if (index_off > len_ || (uint64_t)nc * kIndexEntryBytes > len_ - index_off)
    return fail(VI_ERR_INVALID_DATA, "Failed to read index in shard_%llu", (unsigned long long)shard_id);
  const uint64_t vsz = 4ull * dim_, cpad = pad8(vsz), stride = record_stride(dim_);
  lists_.resize(nc);
  for (uint32_t i = 0; i < nc; ++i) {
    const uint8_t *e = base + index_off + (uint64_t)i * kIndexEntryBytes;
    ShardListView &lv = lists_[i];
    lv.centroid_id = load_le<uint64_t>(e + 0);
    lv.num_vectors = load_le<uint32_t>(e + 8);
    const uint64_t off = load_le<uint64_t>(e + 16), size = load_le<uint64_t>(e + 24);
    if (off > len_ || size > len_ - off || size < vsz)
      return fail(VI_ERR_OTHER, "Failed to read cluster: block out of file bounds");
    // bounds check of every record (shards.rs:310-316)
    const uint64_t need = vsz + cpad + (uint64_t)lv.num_vectors * stride - (lv.num_vectors ? cpad : 0);
    if (need > size)
      return fail(VI_ERR_INVALID_DATA, "Not enough bytes for vector metadata in centroid %llu block",
                  (unsigned long long)lv.centroid_id);
    lv.centroid = base + off;
    lv.records = base + off + vsz + cpad;
  }
  return VI_OK;
}

const ShardListView *ShardFile::find(uint64_t centroid_id) const {
  for (const ShardListView &l : lists_)
    if (l.centroid_id == centroid_id) return &l;
  return nullptr;
}

vi_status shard_save_to(const std::string &shards_dir, uint64_t shard_id, uint32_t dim,
                        uint32_t num_lists, const uint64_t *centroid_ids, const float *centroid_vecs,
                        const uint64_t *list_off, const uint64_t *ids, const uint64_t *ext_ids,
                        const uint64_t *timestamps, const float *vecs) {
  VI_TRY(make_dirs(shards_dir));
  const std::string path = shard_path(shards_dir, shard_id);
  ::unlink(path.c_str());  // shards.rs:73
  const uint64_t vsz = 4ull * dim, pad = pad8(vsz), stride = record_stride(dim);
  const uint64_t data_off = kShardHeaderBytes + kIndexEntryBytes * (uint64_t)num_lists;
  uint64_t total = data_off;
  for (uint32_t i = 0; i < num_lists; ++i) total += vsz + pad + (list_off[i + 1] - list_off[i]) * stride;
  // Assemble the whole image once and write it with a single call.
  std::vector<uint8_t> img(total, 0);
  uint8_t *p = img.data();
  auto put64 = [&](uint64_t off, uint64_t v) { std::memcpy(p + off, &v, 8); };
  auto put32 = [&](uint64_t off, uint32_t v) { std::memcpy(p + off, &v, 4); };
  put64(0, shard_id); put64(8, 1); put32(16, dim); put32(20, num_lists);
  put64(24, kShardHeaderBytes); put64(32, data_off);
  uint64_t cur = data_off;
  for (uint32_t i = 0; i < num_lists; ++i) {
    const uint64_t nv = list_off[i + 1] - list_off[i];
    const uint64_t size = vsz + pad + nv * stride;
    const uint64_t e = kShardHeaderBytes + (uint64_t)i * kIndexEntryBytes;
    put64(e, centroid_ids[i]); put32(e + 8, (uint32_t)nv); put32(e + 12, 0);
    put64(e + 16, cur); put64(e + 24, size);
    std::memcpy(p + cur, centroid_vecs + (uint64_t)i * dim, vsz);
    uint64_t o = cur + vsz + pad;
    for (uint64_t v = list_off[i]; v < list_off[i + 1]; ++v) {
      put64(o, ids[v]); put64(o + 8, ext_ids[v]); put64(o + 16, timestamps[v]);
      std::memcpy(p + o + kVectorMetaBytes, vecs + v * dim, vsz);
      o += stride;
    }
    cur += size;
  }
  FILE *f = fopen(path.c_str(), "wb");
  if (!f) return fail(VI_ERR_IO, "File::create(%s): %s", path.c_str(), strerror(errno));
  const bool ok = img.empty() || fwrite(img.data(), 1, img.size(), f) == img.size();
  if (fclose(f) != 0 || !ok) return fail(VI_ERR_IO, "write(%s) failed", path.c_str());
  return VI_OK;
}

// ---- bincode 2 "standard" varint ----------------------------------------------------------
namespace {
void put_varint(std::vector<uint8_t> &b, uint64_t v) {
  auto raw = [&](int n) { for (int i = 0; i < n; ++i) b.push_back((uint8_t)(v >> (8 * i))); };
  if (v < 251) b.push_back((uint8_t)v);
  else if (v <= 0xFFFF) { b.push_back(251); raw(2); }
  else if (v <= 0xFFFFFFFFull) { b.push_back(252); raw(4); }
  else { b.push_back(253); raw(8); }
}
struct Reader {
  const uint8_t *p; size_t len, off = 0;
  bool varint(uint64_t *v) {
    if (off >= len) return false;
    const uint8_t t = p[off++];
    int nb;
    if (t < 251) { *v = t; return true; }
    if (t == 251) nb = 2; else if (t == 252) nb = 4; else if (t == 253) nb = 8; else return false;
    if (off + nb > len) return false;
    uint64_t x = 0;
    for (int i = 0; i < nb; ++i) x |= (uint64_t)p[off + i] << (8 * i);
    off += nb;
    *v = x;
    return true;
  }
  bool byte(uint8_t *v) { if (off >= len) return false; *v = p[off++]; return true; }
  bool floats(float *dst, uint64_t n) {
    if (n > (len - off) / 4) return false;
    std::memcpy(dst, p + off, n * 4);
    off += n * 4;
    return true;
  }
};
vi_status slurp(const std::string &path, std::vector<uint8_t> *out, vi_status not_found) {
  FILE *f = fopen(path.c_str(), "rb");
  if (!f) return fail(errno == ENOENT ? not_found : VI_ERR_IO, "open(%s): %s", path.c_str(), strerror(errno));
  fseek(f, 0, SEEK_END);
  long sz = ftell(f);
  fseek(f, 0, SEEK_SET);
  out->resize(sz > 0 ? (size_t)sz : 0);
  const bool ok = out->empty() || fread(out->data(), 1, out->size(), f) == out->size();
  fclose(f);
  return ok ? VI_OK : fail(VI_ERR_IO, "read(%s) failed", path.c_str());
}
}  // namespace

vi_status index_meta_save(const IndexMeta &m, const std::string &index_dir) {
  VI_TRY(make_dirs(index_dir));
  std::vector<uint8_t> b;
  const uint64_t k = m.k(), d = m.dimension;
  b.reserve(k * (d * 4 + 12) + 64);
  b.push_back(1); put_varint(b, k); put_varint(b, k);       // ndarray: version, dim, data len
  for (uint64_t c = 0; c < k; ++c) {
    put_varint(b, c); put_varint(b, d);                       // Centroid{id, vector}
    const uint8_t *src = (const uint8_t *)(m.centroids.data() + c * d);
    b.insert(b.end(), src, src + d * 4);
  }
  b.push_back(1); put_varint(b, k); put_varint(b, k);
  for (uint64_t c = 0; c < k; ++c) put_varint(b, m.c2s[c]);
  put_varint(b, d);
  const std::string path = index_dir + "/index.bin";
  FILE *f = fopen(path.c_str(), "wb");
  if (!f) return fail(VI_ERR_IO, "File::create(%s): %s", path.c_str(), strerror(errno));
  const bool ok = fwrite(b.data(), 1, b.size(), f) == b.size();
  if (fclose(f) != 0 || !ok) return fail(VI_ERR_IO, "write(%s) failed", path.c_str());
  return VI_OK;
}

vi_status index_meta_load(const std::string &index_dir, IndexMeta *out) {
  std::vector<uint8_t> raw;
  VI_TRY(slurp(index_dir + "/index.bin", &raw, VI_ERR_NOT_FOUND));
  Reader r{raw.data(), raw.size()};
  auto bad = [] { return fail(VI_ERR_OTHER, "Bincode decoding error: malformed index.bin"); };
  uint8_t ver;
  uint64_t k, k2, v;
  if (!r.byte(&ver) || ver != 1 || !r.varint(&k) || !r.varint(&k2) || k != k2) return bad();
  uint64_t d0 = 0;
  out->centroids.clear();
  for (uint64_t c = 0; c < k; ++c) {
    uint64_t id, dl;
    if (!r.varint(&id) || !r.varint(&dl)) return bad();
    if (c == 0) { d0 = dl; if (dl && k > (raw.size() / 4) / dl + 1) return bad(); out->centroids.resize(k * dl); }
    if (dl != d0 || !r.floats(out->centroids.data() + c * d0, dl)) return bad();
  }
  if (!r.byte(&ver) || ver != 1 || !r.varint(&v) || v != k || !r.varint(&v) || v != k) return bad();
  out->c2s.resize(k);
  for (uint64_t c = 0; c < k; ++c)
    if (!r.varint(&out->c2s[c])) return bad();
  if (!r.varint(&v)) return bad();
  out->dimension = (uint32_t)v;
  if (k > 0 && d0 != out->dimension) return bad();
  return VI_OK;
}

vi_status read_vectors_from_file(const std::string &path, std::vector<VectorFileRecord> *out) {
  std::vector<uint8_t> raw;
  VI_TRY(slurp(path, &raw, VI_ERR_OTHER));  // api.rs:156 wraps every failure as Other
  Reader r{raw.data(), raw.size()};
  out->clear();
  while (r.off < r.len) {
    const size_t batch_start_count = out->size();
    uint64_t cnt;
    bool ok = r.varint(&cnt) && cnt <= r.len - r.off;  // every record takes at least three bytes
    for (uint64_t i = 0; ok && i < cnt; ++i) {
      VectorFileRecord rec;
      uint64_t vl;
      ok = r.varint(&rec.id) && r.varint(&vl) && vl <= (r.len - r.off) / 4;  // (vl * 4 would wrap for vl >= 2^62)
      if (ok) { rec.values.resize(vl); ok = r.floats(rec.values.data(), vl) && r.varint(&rec.meta); }
      if (ok) out->push_back(std::move(rec));
    }
    if (!ok) { out->resize(batch_start_count); break; }  // Err(_) => break (utils.rs:102)
  }
  return VI_OK;
}

}  // namespace vi
