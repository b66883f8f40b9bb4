"""CPU: the oracle (oracle/vi_oracle.c) against the committed golden fixtures.

Fixtures come from tests/golden/make_golden.py (independent numpy/struct
implementation written from the reference's source; the reference itself is a
Rust crate that cannot be built in this image)."""
import ctypes as C
import hashlib
import json
import os

import numpy as np
import pytest

import oracle_lib as O

G = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    with open(os.path.join(G, name)) as f:
        return json.load(f)


def from_bits(bits):
    return np.array(bits, dtype=np.uint32).view(np.float32)


def bits_of(x):
    return int(np.array([x], dtype=np.float32).view(np.uint32)[0])


def test_l2sq_known_answers_bit_exact():
    for kat in load("l2sq.json"):
        a, b = from_bits(kat["a_bits"]), from_bits(kat["b_bits"])
        assert bits_of(O.l2sq_scalar(a, b)) == kat["scalar_bits"], kat["d"]
        assert bits_of(O.l2sq_simd(a, b)) == kat["lanes_bits"], kat["d"]


def test_heuristic_tables():
    L = O.lib()
    for r in load("heuristics.json"):
        n = r["n"]
        assert L.orc_calculate_num_clusters(n) == r["k"]
        assert L.orc_calculate_max_iterations(n) == r["max_iters"]
        assert L.orc_minibatch_size(n) == r["batch"]
        assert L.orc_meta_k(max(r["k"], 1)) == r["meta_k"]
        assert L.orc_num_shards(r["k"]) == r["num_shards"]


def _dataset(name, n, d):
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(G, "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    return mg.make_records(d, n) if name == "make_records" else mg.create_test_vectors(n, d)


@pytest.mark.parametrize("case", load("exhaustive.json"), ids=lambda c: c["name"])
def test_exhaustive_probe_equals_brute_force(case, tmp_path):
    """n_probe >= #lists makes search independent of k-means (tests/api_tests.rs:40-92)."""
    X = _dataset(case["data"], case["n"], case["d"])
    ix = O.OracleIndex.build(X, str(tmp_path / "index"), str(tmp_path / "shards"))
    q = from_bits(case["query_bits"])
    rc, ids, ds, _ = ix.search(q, case["k"], 10_000)
    assert rc == O.ORC_OK
    assert [bits_of(x) for x in ds] == case["dist_bits"]
    if not case["has_ties"]:
        assert ids.tolist() == case["ids"]
    # reload from disk: same answer (index.bin + shard files round trip)
    ix2 = O.OracleIndex.load(str(tmp_path / "index"), str(tmp_path / "shards"))
    rc, ids2, ds2, _ = ix2.search(q, case["k"], 10_000)
    assert ids2.tolist() == ids.tolist() and ds2.tobytes() == ds.tobytes()


@pytest.mark.parametrize("sc", load("shards.json"), ids=lambda s: s["name"])
def test_shard_file_bytes(sc, tmp_path):
    lists = [(np.array([v["id"] for v in l["vectors"]], dtype=np.uint64),
              np.array([v["ext"] for v in l["vectors"]], dtype=np.uint64),
              np.array([v["ts"] for v in l["vectors"]], dtype=np.uint64),
              np.array([v["v"] for v in l["vectors"]], dtype=np.float32).reshape(len(l["vectors"]), sc["dim"]))
             for l in sc["lists"]]
    cids = [l["centroid_id"] for l in sc["lists"]]
    cv = np.array([l["centroid"] for l in sc["lists"]], dtype=np.float32).reshape(len(cids), sc["dim"])
    rc = O.shard_save(str(tmp_path), sc["shard_id"], sc["dim"], cids, cv, lists)
    assert rc == O.ORC_OK
    raw = open(tmp_path / f"shard_{sc['shard_id']}.bin", "rb").read()
    assert len(raw) == sc["size"]
    assert raw.hex() == sc["hex"]
    # reader: write the golden bytes and read them back selectively, reversed order
    os.makedirs(tmp_path / "g", exist_ok=True)
    open(tmp_path / "g" / f"shard_{sc['shard_id']}.bin", "wb").write(bytes.fromhex(sc["hex"]))
    rc, got = O.shard_get(str(tmp_path / "g"), sc["shard_id"], list(reversed(cids)))
    assert rc == O.ORC_OK
    for (cid, cent, metas, vecs), l in zip(got, reversed(sc["lists"])):
        assert cid == l["centroid_id"]
        assert cent.tolist() == np.array(l["centroid"], dtype=np.float32).tolist()
        assert metas.tolist() == [[v["id"], v["ext"], v["ts"]] for v in l["vectors"]]
        assert vecs.tobytes() == np.array([v["v"] for v in l["vectors"]], dtype=np.float32).tobytes()


def test_shard_reader_errors(tmp_path):
    sc = load("shards.json")[0]
    # missing file -> Err (tests/shards_tests.rs:541-554)
    rc, _ = O.shard_get(str(tmp_path), 999, [1])
    assert rc == O.ORC_OTHER
    open(tmp_path / f"shard_{sc['shard_id']}.bin", "wb").write(bytes.fromhex(sc["hex"]))
    # unknown centroid -> NotFound (:558-584)
    rc, _ = O.shard_get(str(tmp_path), sc["shard_id"], [12345])
    assert rc == O.ORC_NOT_FOUND
    # corrupt header: first 4 bytes 0xFF -> shard id mismatch -> Err (:588-630)
    raw = bytearray(bytes.fromhex(sc["hex"]))
    raw[0:4] = b"\xff\xff\xff\xff"
    open(tmp_path / f"shard_{sc['shard_id']}.bin", "wb").write(bytes(raw))
    rc, _ = O.shard_get(str(tmp_path), sc["shard_id"], [5])
    assert rc == O.ORC_INVALID_DATA


def test_index_bin_bytes(tmp_path):
    tiny = load("index_bin.json")[0]
    # build a matching index through the oracle is k-means dependent; instead check the
    # loader on the golden bytes (writer is covered by the exhaustive round trips)
    os.makedirs(tmp_path / "index")
    open(tmp_path / "index" / "index.bin", "wb").write(bytes.fromhex(tiny["hex"]))
    ix = O.OracleIndex.load(str(tmp_path / "index"), str(tmp_path / "shards"))
    Cn, c2s = ix.centroids()
    assert ix.dimension == 3 and ix.num_centroids == 3
    assert Cn.tobytes() == np.array(tiny["C"], dtype=np.float32).tobytes()
    assert c2s.tolist() == tiny["c2s"]


def test_rng_stream_matches_python_restatement():
    L = O.lib()
    for g in load("rng.json"):
        r = O.OrcRng()
        L.orc_rng_seed_from_u64(C.byref(r), g["seed"])
        assert [L.orc_rng_next_u32(C.byref(r)) for _ in range(5)] == g["u32"]
        assert [L.orc_rng_next_u64(C.byref(r)) for _ in range(40)] == g["u64"]
        L.orc_rng_seed_from_u64(C.byref(r), g["seed"])
        got = [L.orc_rng_gen_range_usize(C.byref(r), 0, n) for n in [1, 2, 10, 150, 50_000, 10 ** 7, 2 ** 40 + 3]]
        assert got == g["gen_range"]
        L.orc_rng_seed_from_u64(C.byref(r), g["seed"])
        perm = np.arange(37, dtype=np.uint64)
        L.orc_rng_shuffle_u64(C.byref(r), O._p(perm), 37)
        assert perm.tolist() == g["shuffle37"]


def test_chacha_core_rfc7539_vector():
    """RFC 7539 §2.3.2 (20 rounds) pins the quarter-round/permutation; StdRng uses 12."""
    key = np.arange(32, dtype=np.uint8).view("<u4").copy()
    out = np.zeros(16, dtype=np.uint32)
    O.lib().orc_chacha_block(O._p(key), 1 | (0x09000000 << 32), 0x4A000000, 20, O._p(out))
    assert [hex(x) for x in out[:4]] == ["0xe4e7f110", "0x15593bd1", "0x1fdd0f50", "0xc47120a3"]
    assert hex(out[15]) == "0x4e3c50a2"


def test_bench_dataset_checksum():
    sha = load("dataset_sha.json")
    rng = np.random.default_rng(42)
    xb = rng.standard_normal((50000, 64)).astype(np.float32)
    assert hashlib.sha256(xb.tobytes()).hexdigest() == sha["c1_xb_50000x64_seed42"]
