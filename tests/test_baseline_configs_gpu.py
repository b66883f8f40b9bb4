"""GPU parity at the BASELINE.json configurations themselves (not scaled-down stand-ins), through the C ABI, against
the CPU oracle on the same index files:

  C1  N=50 000  D=64   nlist=100   nprobe=8   k=10   all 1 000 queries; index files byte-identical to the oracle's build
  C2  N=1e6     D=128  nlist=4096  nprobe 16 and 32, k=10, 1 000 queries: ids AND distance bits, every ranking engine
  C3  k-means exact assign N=1e6 x k=16384 x D=128 (the full-size centroid table; N bounded so that the oracle finishes):
      labels vs orc_assign_brute_force on every row the MFMA tiers left undecided + 20 000 sampled rows

  C3  the training loop itself at N=1e7 k=16384 D=128 (sampled k-means++ with 16 384 draws, 20 mini-batch iterations):
      centroid bits vs orc_kmeans_mini_batch; final labels (reference 2-level mode and exact mode) on 20 000 sampled rows
  C4  the C2 index loaded as 8 in-process ranks (one device) under BOTH partition rules, 1 000 queries, nprobe 16 / 32:
      split coarse step + per-rank scan + packed merge == the oracle's single search (an 8-GPU node is not available to
      the tests: the RCCL all-gather is replaced by the ranks writing side by side into one buffer)
  C5  one rank's slice of N=1e8 D=96 nlist=65536 (N=1.25e7, X ~ N(0,1): real-valued => bf16 x 3 ranking, 65 536-row coarse
      table): built by the product, 500 queries at nprobe 32 == the oracle on the same files
"""
import ctypes as C
import filecmp
import os

import numpy as np
import pytest

import oracle_lib as O
import vector_indexer_py as vip
from hiprt import Hip
from vector_indexer_py import _native as N

pytestmark = pytest.mark.gpu

ENGINES = {
    "default": {},
    "bf16x3 (lo planes streamed)": {"VI_FILTER_HI_ONLY": "0"},
    "f32 MFMA": {"VI_FILTER_BF16": "0"},
    "exact-order VALU": {"VI_FILTER": "0"},
    "32-query work items": {"VI_FILTER_GQ": "32"},
    "128-query work items": {"VI_FILTER_GQ": "128"},
}


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def assert_same(Dg, Ig, Do, Io, what):
    bad = np.nonzero((Ig != Io).any(axis=1) | (bits(Dg) != bits(Do)).any(axis=1))[0]
    assert bad.size == 0, (f"{what}: {bad.size} queries differ, first {bad[0]}: gpu {Ig[bad[0]]} {Dg[bad[0]]} "
                           f"oracle {Io[bad[0]]} {Do[bad[0]]}")


# ---------------------------------------------------------------------------------------------------------------
# C2: the headline search configuration at full size
# ---------------------------------------------------------------------------------------------------------------
def sift_shaped(n, d, nq, seed):
    """numpy port of bench.py:make_dataset (same recipe, numpy's generator): non-negative integer-valued f32 in
    [0, 218] drawn from a 2048-component mixture; queries are held-out draws of the same mixture"""
    rng = np.random.default_rng(seed)
    centers = rng.standard_normal((2048, d), dtype=np.float32) * 30.0 + 60.0

    def draw(m):
        out = np.empty((m, d), dtype=np.float32)
        for s in range(0, m, 100_000):
            e = min(m, s + 100_000)
            x = centers[rng.integers(0, 2048, e - s)] + rng.standard_normal((e - s, d), dtype=np.float32) * 14.0
            out[s:e] = np.clip(np.round(np.abs(x)), 0, 218)
        return out
    return draw(n), draw(nq)


@pytest.fixture(scope="module")
def c2(tmp_path_factory):
    xb, xq = sift_shaped(1_000_000, 128, 10_000, 42)
    work = str(tmp_path_factory.mktemp("c2"))
    gpu = vip.build(xb, work, nlist=4096, now_secs=1_700_000_000)
    orc = O.OracleIndex.load(os.path.join(work, "index"), os.path.join(work, "shards"))
    gpu.work_dir = work
    return gpu, orc, xq[:1000].copy(), xq


@pytest.mark.parametrize("n_probe", [16, 32])
def test_c2_sift1m_shape_ids_and_distance_bits(c2, n_probe, monkeypatch):
    """src/ivf_index.rs:190-267 at N=1e6 D=128 nlist=4096 k=10: host-pointer entry, every engine"""
    gpu, orc, Q, _ = c2
    assert gpu.num_vectors == 1_000_000 and gpu.num_centroids == orc.num_centroids
    rc, Do, Io = orc.search_batch(Q, 10, n_probe, O.usable_cpus())
    assert rc == O.ORC_OK
    for name, env in ENGINES.items():
        with monkeypatch.context() as m:
            for k, v in env.items():
                m.setenv(k, v)
            Dg, Ig = gpu.search_sync(Q, 10, n_probe)
            assert_same(Dg, Ig, Do, Io, f"nprobe {n_probe}, engine {name}")


def test_c2_full_batch_device_entry_matches_host_entry(c2):
    """the bench's call (vi_indexer_search_device, 10 000 device-resident queries) returns what the host-pointer call
    returns for the same queries: the first 1 000 rows are the oracle-checked ones"""
    gpu, orc, Q, xq = c2
    nq, k = xq.shape[0], 10
    hip = Hip()
    try:
        xd = hip.upload(xq)
        D, I = hip.alloc(nq * k * 4), hip.alloc(nq * k * 8)
        for n_probe in (16, 32):
            gpu.search_device(xd, nq, k, n_probe, D, I, 0)
            Dh, Ih = hip.download(D, (nq, k), np.float32), hip.download(I, (nq, k), np.int64)
            rc, Do, Io = orc.search_batch(Q, k, n_probe, O.usable_cpus())
            assert_same(Dh[:1000], Ih[:1000], Do, Io, f"device entry nprobe {n_probe}")
            # the other 9 000: internal consistency (ascending, ids valid); the oracle checks a further sample
            assert (np.diff(Dh, axis=1) >= 0).all() and ((Ih >= 0) & (Ih < 1_000_000)).all()
            rows = np.arange(1000, nq, 45)
            rc, Do, Io = orc.search_batch(xq[rows], k, n_probe, O.usable_cpus())
            assert_same(Dh[rows], Ih[rows], Do, Io, f"device entry nprobe {n_probe}, sampled rows")
    finally:
        hip.close()


# ---------------------------------------------------------------------------------------------------------------
# C1: the reference's own CPU-runnable configuration
# ---------------------------------------------------------------------------------------------------------------
def test_c1_build_and_search(tmp_path):
    """N=50 000 D=64 nlist=100 nprobe=8 k=10 (SURVEY 8d recipe: default_rng(42).standard_normal, queries = next draw):
    the GPU build writes the oracle's files byte for byte (exact k-means++ since N <= 50 000, brute-force assign since
    k <= 100), and all 1 000 queries return the oracle's ids and distance bits"""
    rng = np.random.default_rng(42)
    xb = rng.standard_normal((50_000, 64)).astype(np.float32)
    xq = rng.standard_normal((1_000, 64)).astype(np.float32)
    wg, wo = str(tmp_path / "gpu"), str(tmp_path / "orc")
    gpu = vip.build(xb, wg, nlist=100, now_secs=1_700_000_000)
    os.makedirs(wo + "/index"), os.makedirs(wo + "/shards")
    orc = O.OracleIndex.build(xb, wo + "/index", wo + "/shards", nlist=100, now=1_700_000_000)
    assert filecmp.cmp(wg + "/index/index.bin", wo + "/index/index.bin", shallow=False)
    names = sorted(os.listdir(wo + "/shards"))
    assert names == sorted(os.listdir(wg + "/shards")) and len(names) == 10
    for f in names:
        assert filecmp.cmp(f"{wg}/shards/{f}", f"{wo}/shards/{f}", shallow=False), f
    rc, Do, Io = orc.search_batch(xq, 10, 8, O.usable_cpus())
    assert rc == O.ORC_OK
    Dg, Ig = gpu.search_sync(xq, 10, 8)
    assert_same(Dg, Ig, Do, Io, "C1")


# ---------------------------------------------------------------------------------------------------------------
# C3: exact nearest-centroid assignment at the full-size centroid table
# ---------------------------------------------------------------------------------------------------------------
def test_c3_exact_assign_k16384(monkeypatch):
    """VI_ASSIGN_EXACT == assign_points_brute_force (src/kmeans.rs:462-470) at k=16384 D=128: every row the bf16 x 3 tier
    left undecided (they go through the f32 MFMA tier and the exact scan — the two fallback tiers) plus 20 000 sampled
    rows, N=1e6 rows ranked (the oracle needs ~3 s per 30 000 rows; N is bounded by that, k and D are BASELINE's)"""
    n, d, k = 1_000_000, 128, 16384
    rng = np.random.default_rng(42)
    X = rng.standard_normal((n, d), dtype=np.float32)
    Cn = X[rng.choice(n, k, replace=False)].copy()
    # duplicate centroids and exact ties must resolve to the lower index (strict '<', kmeans.rs:364-370)
    Cn[k - 1] = Cn[7]
    Cn[1000] = Cn[999]
    hip = Hip()
    try:
        Xd, Cd = hip.upload(X), hip.upload(Cn)
        lab, amb = hip.alloc(n * 4), hip.alloc(n * 4)
        st = N.AssignStats()
        st.ambiguous_rows_dev = amb
        st.ambiguous_cap = n
        N.check(N.lib().vi_assign_device(0, Xd, n, d, Cd, k, 42, N.VI_ASSIGN_EXACT, lab, C.byref(st)))
        assert st.used_mfma == 1
        n_amb = int(st.tier1_rows)
        assert 0 < n_amb < n // 10 and int(st.ambiguous_rows) <= n_amb
        rows_amb = hip.download(amb, (n,), np.uint32)[:n_amb].astype(np.int64)
        assert (rows_amb < n).all() and np.unique(rows_amb).size == n_amb
        rows = np.unique(np.concatenate([rows_amb, rng.choice(n, 20_000, replace=False), [0, n - 1]]))
        Xh = np.ascontiguousarray(X[rows])
        want = O.assign(Xh, Cn, mode="brute")
        got = hip.download(lab, (n,), np.uint32)[rows].astype(np.uint64)
        bad = np.nonzero(got != want)[0]
        assert bad.size == 0, (f"{bad.size} of {rows.size} labels differ (first row {rows[bad[0]]}: "
                               f"{got[bad[0]]} vs {want[bad[0]]}); tiers: {n_amb} undecided by bf16x3, "
                               f"{int(st.ambiguous_rows)} re-evaluated exactly")
        # the exact-order engine alone (no MFMA tiers) agrees on the same rows
        monkeypatch.setenv("VI_NO_MFMA", "1")
        lab2 = hip.alloc(rows.size * 4)
        N.check(N.lib().vi_assign_device(0, hip.upload(Xh), rows.size, d, Cd, k, 42, N.VI_ASSIGN_EXACT, lab2, None))
        assert (hip.download(lab2, (rows.size,), np.uint32).astype(np.uint64) == want).all()
    finally:
        hip.close()


# ---------------------------------------------------------------------------------------------------------------
# C4: the C2 index over 8 ranks, both partition rules (in-process: one device plays every rank)
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("placement", [0, 1], ids=["stripes", "shard-placement"])
def test_c4_eight_ranks_on_the_c2_index(c2, placement):
    """src/ivf_index.rs:104-164 (lists -> shard files) decides what rank holds what under placement 1; stripes cut every
    list by 64-vector blocks.  The multi-GPU protocol of bench.py --gpus 8: coarse step split by query
    (vi_indexer_probe_device on each rank's slice), the probe lists side by side (the all-gather), every rank's scan of
    the whole batch into its packed [D | I | tie] buffer (vi_indexer_search_probed_device), the packed merge — must return
    the oracle's ids and distance bits (src/ivf_index.rs:190-267) at N=1e6, nlist=4096."""
    gpu, orc, Q, _ = c2
    world, d, k = 8, 128, 10
    idx, sh = os.path.join(gpu.work_dir, "index"), os.path.join(gpu.work_dir, "shards")
    parts = [vip.load(idx, sh, d, rank=r, world_size=world, placement=placement) for r in range(world)]
    resident = [p.num_vectors for p in parts]
    assert all(p.num_centroids == orc.num_centroids for p in parts)
    if placement == 1:
        assert sum(resident) == 1_000_000 and min(resident) > 0      # every vector on exactly one rank
    else:
        assert sum(resident) == 1_000_000 and max(resident) - min(resident) < 64 * orc.num_centroids
    hip = Hip()
    try:
        nq = Q.shape[0]
        xq = hip.upload(Q)
        S = int(N.lib().vi_packed_result_bytes(nq, k))
        off_i = (nq * k * 4 + 7) // 8 * 8
        packed, Dm, Im = hip.alloc(world * S), hip.alloc(nq * k * 4), hip.alloc(nq * k * 8)
        per = (nq + world - 1) // world
        for n_probe in (16, 32):
            rc, Do, Io = orc.search_batch(Q, k, n_probe, O.usable_cpus())
            assert rc == O.ORC_OK
            probes, order = hip.alloc(nq * n_probe * 4), hip.alloc(nq * n_probe * 4)
            for r, p in enumerate(parts):
                q0, q1 = min(nq, r * per), min(nq, (r + 1) * per)
                assert p.probe_device(xq + q0 * d * 4, q1 - q0, n_probe, probes + q0 * n_probe * 4,
                                      order + q0 * n_probe * 4) == n_probe
            for r, p in enumerate(parts):
                b = packed + r * S
                p.search_probed_device(xq, nq, k, n_probe, probes, order, b, b + off_i, b + off_i + nq * k * 8)
            N.check(N.lib().vi_merge_partials_packed_device(0, nq, k, world, packed, Dm, Im))
            assert_same(hip.download(Dm, (nq, k), np.float32), hip.download(Im, (nq, k), np.int64), Do, Io,
                        f"C4 placement {placement} nprobe {n_probe}, split coarse + packed merge")
            # the unsplit entry (every rank runs its own coarse step) and the three-array merge
            Dg, Ig, Tg = hip.alloc(world * nq * k * 4), hip.alloc(world * nq * k * 8), hip.alloc(world * nq * k * 8)
            for r, p in enumerate(parts):
                p.search_device(xq, nq, k, n_probe, Dg + r * nq * k * 4, Ig + r * nq * k * 8, Tg + r * nq * k * 8)
            N.check(N.lib().vi_merge_partials_device(0, nq, k, world, Dg, Ig, Tg, Dm, Im))
            assert_same(hip.download(Dm, (nq, k), np.float32), hip.download(Im, (nq, k), np.int64), Do, Io,
                        f"C4 placement {placement} nprobe {n_probe}, per-rank coarse")
    finally:
        hip.close()
        del parts


# ---------------------------------------------------------------------------------------------------------------
# C3: the training loop at full size
# ---------------------------------------------------------------------------------------------------------------
def test_c3_mini_batch_train_n1e7_k16384():
    """run_kmeans_mini_batch (src/kmeans.rs:64-150) at BASELINE C3: N=1e7 D=128 k=16384, max_iters=20
    (calculate_max_iterations(1e7), utils.rs:18-26), batch 256: the sampled k-means++ branch (N > 50 000,
    kmeans.rs:232-310: 16 384 draws over a 50 000-row sample), 20 full shuffles + mini-batch blends + re-seeds of the
    clusters never hit.  Centroid bits must equal the oracle's; the final assignment (kmeans.rs:146-147 -> :445-581) is
    checked on 20 000 sampled rows in the reference's 2-level mode and in exact mode (the oracle's own final pass over
    1e7 rows would take minutes; the labels of a row depend on that row and the centroids only)."""
    n, d, k, iters = 10_000_000, 128, 16384, 20
    assert int(N.lib().vi_calculate_max_iterations(n)) == iters and int(N.lib().vi_minibatch_size(n)) == 256
    rng = np.random.default_rng(42)
    X = np.empty((n, d), dtype=np.float32)
    for s in range(0, n, 1_000_000):
        X[s:s + 1_000_000] = rng.standard_normal((1_000_000, d), dtype=np.float32)
    hip = Hip()
    try:
        Xd = hip.upload(X)
        Cd, lab = hip.alloc(k * d * 4), hip.alloc(n * 4)
        it = C.c_uint64(0)
        N.check(N.lib().vi_kmeans_mini_batch_device(0, Xd, n, d, k, iters, -1.0, 42, N.VI_ASSIGN_REFERENCE, Cd, lab,
                                                    C.byref(it)))
        Cg = hip.download(Cd, (k, d), np.float32)
        lab_ref = hip.download(lab, (n,), np.uint32)
        rc, Co, _, it_o = O.kmeans_mini_batch(X, k, iters, None, 42, want_labels=False)
        assert rc == O.ORC_OK and it.value == it_o
        bad = np.nonzero((bits(Cg) != bits(Co)).any(axis=1))[0]
        assert bad.size == 0, f"{bad.size} of {k} centroids differ from the oracle's, first {bad[0]}"
        rows = np.unique(np.concatenate([rng.choice(n, 20_000, replace=False), [0, n - 1]]))
        Xs = np.ascontiguousarray(X[rows])
        want = O.assign(Xs, Co, seed=42, mode="hier")
        got = lab_ref[rows].astype(np.uint64)
        assert (got == want).all(), f"{(got != want).sum()} of {rows.size} reference-mode labels differ"
        # exact mode: same training loop (same centroids), final assignment == brute force
        N.check(N.lib().vi_kmeans_mini_batch_device(0, Xd, n, d, k, iters, -1.0, 42, N.VI_ASSIGN_EXACT, Cd, lab, None))
        assert (bits(hip.download(Cd, (k, d), np.float32)) == bits(Co)).all()
        want = O.assign(Xs, Co, mode="brute")
        got = hip.download(lab, (n,), np.uint32)[rows].astype(np.uint64)
        assert (got == want).all(), f"{(got != want).sum()} of {rows.size} exact-mode labels differ"
    finally:
        hip.close()


# ---------------------------------------------------------------------------------------------------------------
# C5: one rank's slice (1/8 of N=1e8) at the full list count
# ---------------------------------------------------------------------------------------------------------------
def test_c5_per_rank_slice_n12_5m_d96_nlist65536(tmp_path_factory):
    """BASELINE C5 = N=1e8 D=96 nlist=65536 over 8 GPUs; one rank holds N/8 = 1.25e7 vectors.  X ~ N(0,1): real-valued
    lists => the bf16 x 3 ranking arithmetic, a 65 536-row coarse table on the MFMA coarse path.  The product builds the
    index (GPU k-means, GPU list build, shard export), the oracle loads the very same files, and 500 queries at
    nprobe 32 (+ 100 at nprobe 8 and 64) must return its ids and distance bits (src/ivf_index.rs:190-267)."""
    n, d, nlist, k = 12_500_000, 96, 65536, 10
    rng = np.random.default_rng(42)
    xb = np.empty((n, d), dtype=np.float32)
    for s in range(0, n, 1_250_000):
        xb[s:s + 1_250_000] = rng.standard_normal((1_250_000, d), dtype=np.float32)
    xq = rng.standard_normal((500, d), dtype=np.float32)
    work = str(tmp_path_factory.mktemp("c5"))
    gpu = vip.build(xb, work, nlist=nlist, now_secs=1_700_000_000)
    del xb
    try:
        bs = gpu.build_stats()
        assert gpu.num_vectors == n and bs["nlist"] == nlist and bs["shards"] == 256
        orc = O.OracleIndex.load(os.path.join(work, "index"), os.path.join(work, "shards"))
        assert orc.num_centroids == gpu.num_centroids and orc.num_shards <= 256
        for n_probe, nq in ((32, 500), (8, 100), (64, 100)):
            rc, Do, Io = orc.search_batch(xq[:nq], k, n_probe, O.usable_cpus())
            assert rc == O.ORC_OK
            Dg, Ig = gpu.search_sync(xq[:nq], k, n_probe)
            st = gpu.last_stats()
            assert st["rank_mode"] in (2, 4, 5, 6), st                  # real-valued lists on the matrix cores (bf16 x 3, or hi planes + margin), not a fallback
            assert_same(Dg, Ig, Do, Io, f"C5 slice nprobe {n_probe}")
    finally:
        import shutil
        shutil.rmtree(work, ignore_errors=True)
