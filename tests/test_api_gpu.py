"""GPU: the assertions of the reference's tests/api_tests.rs and tests/integration_tests.rs, run through
the C ABI via the Python mirror of src/api.rs (vector_indexer_py.api)."""
import os
import struct
import threading

import numpy as np
import pytest

import vector_indexer_py as vip
from vector_indexer_py import _native as N
from vector_indexer_py.api import SearchRequest, VectorIndexer, VectorIndexerConfig, VectorRecord

pytestmark = pytest.mark.gpu


def make_records(dim, n):  # tests/api_tests.rs:12-25
    return [VectorRecord(i, [np.float32(i) * np.float32(0.01) + np.float32(j) for j in range(dim)], None)
            for i in range(n)]


def cfg_for(tmp_path, dim, **kw):
    c = VectorIndexerConfig.new(dim).with_index_dir(tmp_path / "index").with_shards_dir(tmp_path / "shards")
    for k, v in kw.items():
        setattr(c, k, v)
    return c


def test_config_new_sets_expected_defaults():  # :28-37
    c = VectorIndexerConfig.new(16)
    assert (c.dimension, c.index_dir, c.shards_dir, c.default_k, c.default_n_probe, c.max_k, c.max_n_probe) == \
        (16, "index", "shards", 10, 20, 10_000, 10_000)


def test_build_writes_to_configured_dirs_and_load_uses_them(tmp_path):  # :40-92
    cfg = cfg_for(tmp_path, 8)
    records = make_records(8, 150)
    VectorIndexer.new(cfg).build_from_records(records)
    assert os.path.exists(tmp_path / "index" / "index.bin")
    assert any(f.startswith("shard_") for f in os.listdir(tmp_path / "shards"))
    loaded = VectorIndexer.load(cfg)
    res = loaded.search(SearchRequest(records[42].values, False, 5, 50))  # n_probe >= #lists => exhaustive
    assert len(res) > 0 and res[0].external_id == 42


def test_search_uses_config_defaults_and_overrides(tmp_path):  # :95-161
    cfg = cfg_for(tmp_path, 8)
    ix = VectorIndexer.new(cfg).build_from_records(make_records(8, 200))
    q = make_records(8, 200)[0].values
    assert len(ix.search(ix.search_request(q))) == 10
    assert len(ix.search(ix.search_request(q).with_k(3).with_n_probe(2))) == 3


def test_search_clamps_to_max_k_and_max_n_probe(tmp_path):  # :164-197
    cfg = cfg_for(tmp_path, 8, max_k=3, max_n_probe=1)
    records = make_records(8, 80)
    ix = VectorIndexer.new(cfg).build_from_records(records)
    assert len(ix.search(SearchRequest(records[0].values, False, 10, 999))) == 3


def test_include_vectors_controls_payload(tmp_path):  # :200-249
    cfg = cfg_for(tmp_path, 8)
    records = make_records(8, 60)
    ix = VectorIndexer.new(cfg).build_from_records(records)
    a = ix.search(SearchRequest(records[1].values, False, 3, 10))
    b = ix.search(SearchRequest(records[1].values, True, 3, 10))
    assert all(r.vector is None for r in a)
    assert all(r.vector is not None and len(r.vector) == 8 for r in b)
    assert b[0].external_id == 1 and np.allclose(b[0].vector, records[1].values)


def test_error_kinds(tmp_path):  # :252-341
    with pytest.raises(RuntimeError):
        VectorIndexer.load(cfg_for(tmp_path / "missing", 8))
    with pytest.raises(RuntimeError) as e:
        VectorIndexer.new(cfg_for(tmp_path, 8)).build_from_records([])
    assert e.value.kind == "InvalidInput"
    recs = make_records(8, 20)
    recs[7] = VectorRecord(7, [0.0] * 5, None)
    with pytest.raises(RuntimeError) as e:
        VectorIndexer.new(cfg_for(tmp_path, 8)).build_from_records(recs)
    assert e.value.kind == "InvalidInput" and "vector dimension mismatch at index 7: expected 8, got 5" in str(e.value)
    ix = VectorIndexer.new(cfg_for(tmp_path, 8)).build_from_records(make_records(8, 20))
    with pytest.raises(RuntimeError) as e:
        ix.search(SearchRequest([0.0] * 7, False, 5, 5))
    assert e.value.kind == "InvalidInput"
    for k, p in [(0, 5), (5, 0)]:
        with pytest.raises(RuntimeError) as e:
            ix.search(SearchRequest([0.0] * 8, False, k, p))
        assert e.value.kind == "InvalidInput"


def _varint(v):
    if v < 251:
        return bytes([v])
    if v <= 0xFFFF:
        return bytes([251]) + struct.pack("<H", v)
    if v <= 0xFFFFFFFF:
        return bytes([252]) + struct.pack("<I", v)
    return bytes([253]) + struct.pack("<Q", v)


def test_build_from_vector_file_smoke_and_dimension_validation(tmp_path):  # :344-391
    dim = 8
    # two appended bincode batches of Vec<(u64, Vec<f32>, u64)> (utils.rs:34-107)
    blob = b""
    for lo, hi in [(0, 30), (30, 50)]:
        blob += _varint(hi - lo)
        for i in range(lo, hi):
            blob += _varint(i) + _varint(dim) + np.full(dim, i, dtype="<f4").tobytes() + _varint(0)
    vf = tmp_path / "vectors.bin"
    vf.write_bytes(blob)
    ix = VectorIndexer.new(cfg_for(tmp_path, dim)).build_from_vector_file(vf)
    res = ix.search(SearchRequest([0.0] * dim, False, 5, 10))
    assert len(res) == 5 and res[0].external_id == 0 and res[0].distance == 0.0
    with pytest.raises(RuntimeError):
        VectorIndexer.new(cfg_for(tmp_path / "bad", dim + 1)).build_from_vector_file(vf)
    with pytest.raises(RuntimeError):
        VectorIndexer.new(cfg_for(tmp_path / "bad2", dim)).build_from_vector_file(tmp_path / "nope.bin")


def test_full_pipeline_persistence_and_consistency(tmp_path):  # integration_tests.rs:17-188
    rng = np.random.default_rng(0)
    X = rng.uniform(-10, 10, size=(500, 16)).astype(np.float32)
    recs = [VectorRecord(i, X[i].tolist(), 1000 + i) for i in range(500)]
    cfg = cfg_for(tmp_path, 16)
    built = VectorIndexer.new(cfg).build_from_records(recs)
    loaded = VectorIndexer.load(cfg)
    for ix in (built, loaded):
        r = ix.search(SearchRequest(X[7].tolist(), True, 5, 30))
        assert r[0].external_id == 7 and r[0].distance == 0.0 and np.array_equal(np.float32(r[0].vector), X[7])
        assert all(0 <= x.external_id < 500 and x.distance >= 0 for x in r)
        assert all(r[i].distance <= r[i + 1].distance for i in range(len(r) - 1))
    runs = [[(x.external_id, x.distance) for x in loaded.search(SearchRequest(X[3].tolist(), False, 10, 5))]
            for _ in range(5)]
    assert all(run == runs[0] for run in runs)


def test_recall_quality_on_known_data(tmp_path):  # integration_tests.rs:310-391, ivf_index_tests.rs:465-498
    rng = np.random.default_rng(1)
    centers = rng.uniform(-20, 20, size=(10, 16)).astype(np.float32)
    X = np.concatenate([c + rng.uniform(-0.5, 0.5, size=(100, 16)).astype(np.float32) for c in centers])
    ix = VectorIndexer.new(cfg_for(tmp_path, 16)).build_from_records([VectorRecord(i, X[i].tolist()) for i in range(len(X))])
    def recall(n_probe):
        tot = 0.0
        for qi in range(0, 1000, 50):
            d = ((X - X[qi]) ** 2).sum(1)
            truth = set(np.argsort(d, kind="stable")[:10].tolist())
            got = {r.external_id for r in ix.search(SearchRequest(X[qi].tolist(), False, 10, n_probe))}
            tot += len(truth & got) / 10
        return tot / 20
    r15, r5 = recall(15), recall(5)
    assert r15 >= 0.7 and r15 >= r5


def test_large_dimension_1536(tmp_path):  # ivf_index_tests.rs:661-686
    rng = np.random.default_rng(2)
    X = rng.standard_normal((120, 1536)).astype(np.float32)
    ix = VectorIndexer.new(cfg_for(tmp_path, 1536)).build_from_records([VectorRecord(i, X[i].tolist()) for i in range(120)])
    r = ix.search(SearchRequest(X[5].tolist(), False, 3, 20))
    assert r[0].external_id == 5 and r[0].distance == 0.0


def test_concurrent_searches_on_one_handle(tmp_path):  # ivf_index_tests.rs:768-807
    rng = np.random.default_rng(3)
    X = rng.standard_normal((2000, 32)).astype(np.float32)
    ix = VectorIndexer.new(cfg_for(tmp_path, 32)).build_from_records([VectorRecord(i, X[i].tolist()) for i in range(2000)])
    expect = {i: [r.external_id for r in ix.search(SearchRequest(X[i].tolist(), False, 5, 8))] for i in range(16)}
    errors = []

    def worker(t):
        try:
            for rep in range(10):
                for i in range(t, 16, 4):
                    got = [r.external_id for r in ix.search(SearchRequest(X[i].tolist(), False, 5, 8))]
                    assert got == expect[i]
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    [t.start() for t in threads]
    [t.join() for t in threads]
    assert not errors


def test_single_vector_index(tmp_path):  # ivf_index_tests.rs:369-392
    ix = VectorIndexer.new(cfg_for(tmp_path, 4)).build_from_records([VectorRecord(99, [1.0, 2.0, 3.0, 4.0])])
    r = ix.search(SearchRequest([1.0, 2.0, 3.0, 4.0], False, 5, 5))
    assert len(r) == 1 and r[0].external_id == 99 and r[0].distance == 0.0


def test_faiss_style_harness_drives_the_engine(tmp_path):
    """the reference's bench methodology (bench_all_ivf.py: eval_setting loop over an adapter with a settable .nprobe,
    K=100 as scripts/run_faiss_bench.sh:54 sets it, JSON + Markdown output) on the engine, small size"""
    from vector_indexer_py import harness as H
    xb, xq, gt = H.synthetic_dataset(6000, 32, 50, 100, seed=42)
    res = H.run(xb, xq, gt, 100, [1, 8, 1000], 0.05, work_dir=str(tmp_path / "w"), verbose=False)
    assert res["n"] == 6000 and res["d"] == 32 and res["k"] == 100 and res["nlist"] > 0 and res["build_time_s"] > 0
    rec = [res["search_results"][f"nprobe={p}"]["recalls"] for p in (1, 8, 1000)]
    assert all(set(r) == {1, 10, 100} for r in rec)
    assert rec[2][1] == 1.0 and rec[2][100] == 1.0            # n_probe >= #lists is exhaustive: the true NN is rank 1
    assert rec[0][100] <= rec[1][100] <= rec[2][100]
    assert all(res["search_results"][f"nprobe={p}"]["nrun"] >= 1 for p in (1, 8, 1000))
    H.save_results([res], str(tmp_path / "out"))
    assert os.path.exists(tmp_path / "out" / "faiss_bench_results.json") and os.path.exists(tmp_path / "out" / "faiss_bench_results.md")
    # the adapter's search == the index's own search at that nprobe
    idx = vip.load(str(tmp_path / "w" / "index"), str(tmp_path / "w" / "shards"), 32)
    ad = H.FaissStyleAdapter(idx, 100)
    ad.nprobe = 8
    D1, I1 = ad.search(xq, 10)
    D2, I2 = idx.search_sync(xq, 10, 8)
    assert ad.d == 32 and (I1 == I2).all() and (D1.view(np.uint32) == D2.view(np.uint32)).all()


def test_build_reports_its_phases(tmp_path):
    rng = np.random.default_rng(3)
    X = rng.standard_normal((20000, 24)).astype(np.float32)
    idx = vip.build(X, str(tmp_path), nlist=64)
    st = idx.build_stats()
    assert st["n"] == 20000 and st["nlist"] == 64 and 0 < st["lists"] <= 64 and st["shards"] == 8
    assert st["shard_bytes"] == 20000 * (24 + 24 * 4)
    parts = sum(st[k] for k in ("ms_upload", "ms_kmeans", "ms_group", "ms_super", "ms_export", "ms_index"))
    assert st["ms_total"] > 0 and abs(parts - st["ms_total"]) < 0.05 * st["ms_total"] + 1.0


def test_timing_levels(tmp_path):
    """vi_indexer_enable_timing: 0 no events, 1 one at every phase boundary, 2 around the list-rank kernel only (what
    bench.py's timed region runs with); the results do not depend on it"""
    rng = np.random.default_rng(11)
    X = rng.integers(0, 200, size=(30000, 32)).astype(np.float32)
    idx = vip.build(X, str(tmp_path), nlist=128)
    Q = X[:600].copy()
    ref = idx.search_sync(Q, 10, 8)
    phases = ("ms_total", "ms_coarse", "ms_group", "ms_scan", "ms_merge")
    st = idx.last_stats()
    assert all(st[p] == 0 for p in phases)
    for level, nonzero in ((True, set(phases)), (2, {"ms_scan"}), (False, set()), (1, set(phases))):
        idx.enable_timing(level)
        D, I = idx.search_sync(Q, 10, 8)
        assert (I == ref[1]).all() and D.tobytes() == ref[0].tobytes()
        st = idx.last_stats()
        assert {p for p in phases if st[p] > 0} == nonzero, (level, {p: st[p] for p in phases})


def test_concurrent_batches_of_different_shapes_on_one_handle(tmp_path):
    """8 threads (more than the 4 search contexts of a handle) issue batches of different sizes, k and n_probe — MFMA
    engine, coarse step on the matrix cores, generic path — at the same time; every result equals the one the same call
    returns alone.  (The reference's search is &self: ivf_index_tests.rs:768-807.)"""
    rng = np.random.default_rng(17)
    X = rng.integers(0, 60, size=(30000, 32)).astype(np.float32)
    idx = vip.build(X, str(tmp_path), nlist=1100)
    shapes = [(1, 10, 8), (300, 5, 16), (700, 10, 32), (64, 64, 3), (513, 1, 64), (40, 100, 70), (1000, 10, 4), (257, 20, 20)]
    Q = [np.ascontiguousarray(rng.integers(0, 60, size=(nq, 32)).astype(np.float32)) for nq, _, _ in shapes]
    want = [idx.search_sync(Q[i], k, p) for i, (_, k, p) in enumerate(shapes)]
    errors = []

    def worker(i):
        try:
            _, k, p = shapes[i]
            for _ in range(6):
                D, I = idx.search_sync(Q[i], k, p)
                assert (I == want[i][1]).all() and (D.view(np.uint32) == want[i][0].view(np.uint32)).all(), shapes[i]
        except BaseException as e:  # noqa: BLE001
            errors.append(e)
    ts = [threading.Thread(target=worker, args=(i,)) for i in range(len(shapes))]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errors, errors[0]
    assert idx.last_stats()["nq"] in {s[0] for s in shapes}
