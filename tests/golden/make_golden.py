#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/.

The reference (NirajNair/vector-indexer) is a Rust crate; this image has no
cargo/rustc, so fixtures cannot be produced by running it.  They are produced
here by an INDEPENDENT numpy-float32 / struct implementation written from the
reference's source text (citations into /root/reference):

  * l2sq.json         squared-L2 known answers in the two summation orders the
                      reference uses (src/utils.rs:28-30 scalar;
                      src/kmeans.rs:377-419 lane-structured)
  * heuristics.json   calculate_num_clusters / calculate_max_iterations /
                      batch size / meta_k / num_shards (src/utils.rs:9-26,
                      src/kmeans.rs:83,483, src/ivf_index.rs:104)
  * exhaustive.json   exhaustive-probe search == brute force top-k on the
                      datasets the reference's tests use
                      (tests/api_tests.rs:12-25,40-92; tests/test_utils/mod.rs:10-16;
                      tests/ivf_index_tests.rs queries)
  * shards.json       hex dumps of shard_<id>.bin for the scenarios of
                      tests/shards_tests.rs, derived from the repr(C) layout in
                      src/shards.rs:22-51,68-177
  * index_bin.json    index/index.bin bytes for a tiny index (bincode 2
                      standard config + ndarray serde; PARITY UNPINNED)
  * rng.json          rand-0.8.5 StdRng stream restated in pure Python
                      (PARITY UNPINNED against the crate; pins C vs Python)
  * dataset_sha.json  sha256 of the synthetic bench datasets (bench recipe
                      bench/faiss_bench_official/bench_all_ivf.py:67-69)

Run:  python tests/golden/make_golden.py      (numpy only; no reference import)
"""
import hashlib
import json
import math
import os
import struct

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
f32 = np.float32


# ----------------------------------------------------------------------------
# distances
# ----------------------------------------------------------------------------
def l2sq_scalar(a, b):
    acc = f32(0)
    for x, y in zip(a, b):
        t = f32(f32(x) - f32(y))
        acc = f32(acc + f32(t * t))
    return acc


def l2sq_lanes(p, c):
    d = len(p)
    a8 = [f32(0)] * 8
    a4 = [f32(0)] * 4
    j = 0
    while j + 8 <= d:
        for l in range(8):
            t = f32(f32(p[j + l]) - f32(c[j + l]))
            a8[l] = f32(a8[l] + f32(t * t))
        j += 8
    while j + 4 <= d:
        for l in range(4):
            t = f32(f32(p[j + l]) - f32(c[j + l]))
            a4[l] = f32(a4[l] + f32(t * t))
        j += 4
    tail = f32(0)
    while j < d:
        t = f32(f32(p[j]) - f32(c[j]))
        tail = f32(tail + f32(t * t))
        j += 1
    r4 = lambda v: f32(f32(f32(v[0] + v[1]) + v[2]) + v[3])
    r8 = f32(r4(a8[:4]) + r4(a8[4:]))
    return f32(f32(r8 + r4(a4)) + tail)


def bits(x):
    return int(np.array([x], dtype=np.float32).view(np.uint32)[0])


def gen_l2sq():
    rng = np.random.default_rng(20240601)
    out = []
    for d in [1, 3, 4, 7, 8, 12, 16, 31, 64, 96, 100, 128, 1536]:
        for rep in range(3):
            scale = [1.0, 10.0, 255.0][rep]
            a = (rng.standard_normal(d) * scale).astype(np.float32)
            b = (rng.standard_normal(d) * scale).astype(np.float32)
            out.append({
                "d": d,
                "a_bits": [bits(x) for x in a],
                "b_bits": [bits(x) for x in b],
                "scalar_bits": bits(l2sq_scalar(a, b)),
                "lanes_bits": bits(l2sq_lanes(a, b)),
            })
    return out


# ----------------------------------------------------------------------------
# heuristics
# ----------------------------------------------------------------------------
def num_clusters(n):
    if n < 10_000:
        return int(math.sqrt(n))
    if n < 100_000:
        return 2 * math.ceil(math.sqrt(n))
    return 4 * math.ceil(math.sqrt(n))


def max_iterations(n):
    return 300 if n < 10_000 else 100 if n < 100_000 else 50 if n < 1_000_000 else 20


def batch_size(n):
    return min(256, max(10, int(np.sqrt(f32(n)))))


def meta_k(k):
    return min(max(int(np.sqrt(f32(k))), 2), k // 2)


def num_shards(k):
    return int(np.ceil(np.sqrt(f32(k))))


def gen_heuristics():
    rows = []
    for n in [1, 2, 10, 50, 100, 150, 200, 2000, 5000, 9_999, 10_000, 50_000, 99_999, 100_000, 1_000_000,
              10_000_000, 100_000_000]:
        k = num_clusters(n)
        rows.append({"n": n, "k": k, "max_iters": max_iterations(n), "batch": batch_size(n),
                     "meta_k": meta_k(max(k, 1)), "num_shards": num_shards(k)})
    return rows


# ----------------------------------------------------------------------------
# exhaustive-probe == brute force
# ----------------------------------------------------------------------------
def make_records(dim, n):
    # tests/api_tests.rs:12-25  v[i][j] = (i as f32)*0.01 + (j as f32)
    i = np.arange(n, dtype=np.float32)[:, None]
    j = np.arange(dim, dtype=np.float32)[None, :]
    return (f32(0.01) * i + j).astype(np.float32)


def create_test_vectors(n, dim):
    # tests/test_utils/mod.rs:10-16  (x as f32 * 0.1) % 50.0
    x = np.arange(n * dim, dtype=np.float32)
    return np.fmod(x * f32(0.1), f32(50.0)).astype(np.float32).reshape(n, dim)


def brute_topk(X, q, k):
    d = np.array([l2sq_scalar(q, row) for row in X], dtype=np.float32)
    order = np.argsort(d, kind="stable")
    return order[:k], d[order[:k]], d


def gen_exhaustive():
    cases = []

    def add(name, X, q, k):
        ids, ds, alld = brute_topk(X, q, k)
        cases.append({"name": name, "n": int(X.shape[0]), "d": int(X.shape[1]),
                      "data": name.split(":")[0], "query_bits": [bits(x) for x in q], "k": k,
                      "ids": [int(i) for i in ids], "dist_bits": [bits(x) for x in ds],
                      # ties make the order depend on the candidate order, i.e. on the index
                      "has_ties": bool(len(set(bits(x) for x in alld[np.argsort(alld, kind='stable')[:k + 1]])) < min(k + 1, len(alld)))})

    X = make_records(8, 150)
    add("make_records:api_42", X, X[42], 10)
    add("make_records:api_42_k1", X, X[42], 1)
    for (n, d) in [(100, 4), (200, 8), (50, 8)]:
        X = create_test_vectors(n, d)
        add(f"create_test_vectors:{n}x{d}:row0", X, X[0], 5)
        add(f"create_test_vectors:{n}x{d}:ones", X, np.full(d, 1.0, dtype=np.float32), 10)
        add(f"create_test_vectors:{n}x{d}:fives", X, np.full(d, 5.0, dtype=np.float32), 10)
    return cases


# ----------------------------------------------------------------------------
# shard file bytes (src/shards.rs:22-51,68-177)
# ----------------------------------------------------------------------------
def shard_bytes(shard_id, dim, lists):
    """lists: [(centroid_id, centroid_vec, [(id, ext, ts, vec), ...]), ...]"""
    vsz = 4 * dim
    pad = (8 - vsz % 8) % 8
    header_size, entry_size, meta_size = 40, 32, 24
    data_off = header_size + entry_size * len(lists)
    hdr = struct.pack("<QQIIQQ", shard_id, 1, dim, len(lists), header_size, data_off)
    idx, blocks, cur = b"", b"", data_off
    for cid, cvec, vecs in lists:
        size = vsz + pad + len(vecs) * (meta_size + vsz + pad)
        idx += struct.pack("<QIIQQ", cid, len(vecs), 0, cur, size)
        cur += size
        blk = np.asarray(cvec, dtype="<f4").tobytes() + b"\0" * pad
        for (vid, ext, ts, v) in vecs:
            blk += struct.pack("<QQQ", vid, ext, ts) + np.asarray(v, dtype="<f4").tobytes() + b"\0" * pad
        blocks += blk
    return hdr + idx + blocks


def gen_shards():
    sc = []

    def add(name, shard_id, dim, lists):
        b = shard_bytes(shard_id, dim, lists)
        sc.append({"name": name, "shard_id": shard_id, "dim": dim,
                   "lists": [{"centroid_id": cid, "centroid": [float(f32(x)) for x in cv],
                              "vectors": [{"id": vid, "ext": ext, "ts": ts, "v": [float(f32(x)) for x in v]}
                                          for (vid, ext, ts, v) in vecs]} for cid, cv, vecs in lists],
                   "size": len(b), "hex": b.hex()})

    # tests/shards_tests.rs:41-90  (1 list x 3 vectors, D=3 -> 208 bytes)
    add("roundtrip_single_centroid", 2000, 3,
        [(5, [10.0, 20.0, 30.0], [(i, i, 0, v) for i, v in
                                   enumerate([[10.1, 20.1, 30.1], [10.2, 20.2, 30.2], [10.3, 20.3, 30.3]])])])
    # :92-160 three lists, D=2
    add("roundtrip_multiple_centroids", 2001, 2,
        [(10, [1.0, 2.0], [(0, 0, 0, [1.1, 2.1])]),
         (11, [5.0, 6.0], [(0, 0, 0, [5.1, 6.1]), (1, 1, 0, [5.2, 6.2])]),
         (12, [9.0, 10.0], [(0, 0, 0, [9.1, 10.1])])])
    # :211-235 empty list
    add("empty_list", 2003, 4, [(7, [0.5, 1.5, 2.5, 3.5], [])])
    # :507-533 huge ids
    big = 2 ** 64 - 1 - 1000
    add("huge_ids", 2010, 3, [(big, [1.0, 2.0, 3.0], [(big + 1, big + 2, big + 3, [1.5, 2.5, 3.5])])])
    # :358-408 metadata preserved
    add("metadata", 2006, 2, [(1, [0.0, 0.0], [(100, 1000, 1234567890, [0.1, 0.2]),
                                                (200, 2000, 1234567891, [0.3, 0.4]),
                                                (300, 3000, 1234567892, [0.5, 0.6])])])
    # empty shard (fit_with_paths writes shards with zero lists, ivf_index.rs:118-120)
    add("no_lists", 3, 8, [])
    return sc


# ----------------------------------------------------------------------------
# index.bin (bincode 2 standard; PARITY UNPINNED)
# ----------------------------------------------------------------------------
def varint(v):
    if v < 251:
        return bytes([v])
    if v <= 0xFFFF:
        return bytes([251]) + struct.pack("<H", v)
    if v <= 0xFFFFFFFF:
        return bytes([252]) + struct.pack("<I", v)
    return bytes([253]) + struct.pack("<Q", v)


def index_bin(C, c2s, dim):
    k = len(C)
    b = b"\x01" + varint(k) + varint(k)
    for i, c in enumerate(C):
        b += varint(i) + varint(len(c)) + np.asarray(c, dtype="<f4").tobytes()
    b += b"\x01" + varint(k) + varint(k)
    for s in c2s:
        b += varint(int(s))
    return b + varint(dim)


def gen_index_bin():
    C = [[0.5, -1.25, 3.0], [100.0, 200.0, 300.0], [1e-3, 0.0, -0.0]]
    c2s = [1, 0, 1]
    small = index_bin(C, c2s, 3)
    # k >= 251 exercises the u16 varint marker
    k = 300
    Cb = (np.arange(k * 2, dtype=np.float32).reshape(k, 2) * f32(0.5)).tolist()
    c2sb = [i % 18 for i in range(k)]
    big = index_bin(Cb, c2sb, 2)
    return [{"name": "tiny", "dim": 3, "C": C, "c2s": c2s, "hex": small.hex()},
            {"name": "k300", "dim": 2, "k": k, "sha256": hashlib.sha256(big).hexdigest(), "size": len(big)}]


# ----------------------------------------------------------------------------
# rand 0.8.5 StdRng restated in pure Python (PARITY UNPINNED vs the crate)
# ----------------------------------------------------------------------------
M32 = 0xFFFFFFFF


def _rotl(x, n):
    return ((x << n) | (x >> (32 - n))) & M32


def chacha_block(key, counter, rounds):
    s = [0x61707865, 0x3320646E, 0x79622D32, 0x6B206574] + list(key) + [counter & M32, counter >> 32, 0, 0]
    x = list(s)

    def qr(a, b, c, d):
        x[a] = (x[a] + x[b]) & M32; x[d] = _rotl(x[d] ^ x[a], 16)
        x[c] = (x[c] + x[d]) & M32; x[b] = _rotl(x[b] ^ x[c], 12)
        x[a] = (x[a] + x[b]) & M32; x[d] = _rotl(x[d] ^ x[a], 8)
        x[c] = (x[c] + x[d]) & M32; x[b] = _rotl(x[b] ^ x[c], 7)

    for _ in range(rounds // 2):
        qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15)
        qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14)
    return [(a + b) & M32 for a, b in zip(x, s)]


class StdRng:
    def __init__(self, seed_u64):
        MUL, INC, M64 = 6364136223846793005, 11634580027462260723, (1 << 64) - 1
        st, key = seed_u64, []
        for _ in range(8):
            st = (st * MUL + INC) & M64
            xs = (((st >> 18) ^ st) >> 27) & M32
            rot = st >> 59
            key.append(((xs >> rot) | (xs << ((32 - rot) & 31))) & M32)
        self.key, self.counter, self.buf, self.idx = key, 0, [0] * 64, 64

    def _refill(self):
        self.buf = []
        for b in range(4):
            self.buf += chacha_block(self.key, self.counter + b, 12)
        self.counter += 4

    def next_u32(self):
        if self.idx >= 64:
            self._refill(); self.idx = 0
        v = self.buf[self.idx]; self.idx += 1
        return v

    def next_u64(self):
        i = self.idx
        if i < 63:
            self.idx += 2
            return (self.buf[i + 1] << 32) | self.buf[i]
        if i >= 64:
            self._refill(); self.idx = 2
            return (self.buf[1] << 32) | self.buf[0]
        x = self.buf[63]
        self._refill(); self.idx = 1
        return (self.buf[0] << 32) | x

    def gen_range_usize(self, lo, hi):
        rng_ = hi - lo
        zone = ((rng_ << (64 - rng_.bit_length())) & ((1 << 64) - 1)) - 1
        while True:
            m = self.next_u64() * rng_
            if (m & ((1 << 64) - 1)) <= zone:
                return lo + (m >> 64)

    def gen_index(self, ubound):
        if ubound <= M32:
            zone = ((ubound << (32 - ubound.bit_length())) & M32) - 1
            while True:
                m = self.next_u32() * ubound
                if (m & M32) <= zone:
                    return m >> 32
        return self.gen_range_usize(0, ubound)

    def shuffle(self, v):
        for i in range(len(v) - 1, 0, -1):
            j = self.gen_index(i + 1)
            v[i], v[j] = v[j], v[i]


def gen_rng():
    out = []
    for seed in [0, 42, 42 * 31 + 7, (42 * 17 + 42), 2 ** 63 + 12345]:
        r = StdRng(seed)
        u32s = [r.next_u32() for _ in range(5)]
        u64s = [r.next_u64() for _ in range(40)]  # crosses the 64-word buffer (odd index straddle)
        r2 = StdRng(seed)
        ranges = [r2.gen_range_usize(0, n) for n in [1, 2, 10, 150, 50_000, 10 ** 7, 2 ** 40 + 3]]
        r3 = StdRng(seed)
        perm = list(range(37))
        r3.shuffle(perm)
        out.append({"seed": seed, "u32": u32s, "u64": u64s, "gen_range": ranges, "shuffle37": perm})
    return out


# ----------------------------------------------------------------------------
# dataset checksums (bench recipe)
# ----------------------------------------------------------------------------
def gen_dataset_sha():
    out = {}
    rng = np.random.default_rng(42)
    xb = rng.standard_normal((50000, 64)).astype(np.float32)
    xq = rng.standard_normal((1000, 64)).astype(np.float32)
    out["c1_xb_50000x64_seed42"] = hashlib.sha256(xb.tobytes()).hexdigest()
    out["c1_xq_1000x64_seed42"] = hashlib.sha256(xq.tobytes()).hexdigest()
    return out


def main():
    def dump(name, obj):
        with open(os.path.join(HERE, name), "w") as f:
            json.dump(obj, f, separators=(",", ":"))
        print(name, os.path.getsize(os.path.join(HERE, name)), "bytes")

    dump("l2sq.json", gen_l2sq())
    dump("heuristics.json", gen_heuristics())
    dump("exhaustive.json", gen_exhaustive())
    dump("shards.json", gen_shards())
    dump("index_bin.json", gen_index_bin())
    dump("rng.json", gen_rng())
    dump("dataset_sha.json", gen_dataset_sha())


if __name__ == "__main__":
    main()
