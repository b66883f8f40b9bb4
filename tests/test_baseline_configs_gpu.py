"""GPU parity at the BASELINE.json configurations themselves (not scaled-down stand-ins), through the C ABI, against
the CPU oracle on the same index files:

  C1  N=50 000  D=64   nlist=100   nprobe=8   k=10   all 1 000 queries; index files byte-identical to the oracle's build
  C2  N=1e6     D=128  nlist=4096  nprobe 16 and 32, k=10, 1 000 queries: ids AND distance bits, every ranking engine
  C3  k-means exact assign N=1e6 x k=16384 x D=128 (the full-size centroid table; N bounded so that the oracle finishes):
      labels vs orc_assign_brute_force on every row the MFMA tiers left undecided + 20 000 sampled rows

C4/C5 need an 8-GPU node; their per-rank code path is covered by test_search_gpu.py (striped ranks) and the gloo tests.
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
