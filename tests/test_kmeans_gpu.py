"""GPU parity: k-means path (assign / k-means++ / mini-batch / Lloyd / index build) through the C ABI
against the CPU oracle.  Both sides draw the same rand-0.8.5 stream on the host, and every distance,
mean and blend is computed in the same f32 order, so centroids and labels are compared BIT-EXACTLY.
(The rand / wide crates themselves are not vendored in the reference: "parity unpinned" for the
stream and the f32x4 reduce order, see oracle/vi_oracle.h.)"""
import filecmp
import os

import numpy as np
import pytest

import oracle_lib as O
import vector_indexer_py as vip
from vector_indexer_py import _native as N

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def gaussian_clusters(rng, nc, per, d, sep):
    # tests/test_utils/mod.rs:34-66 (uniform +-0.5 noise around separated centres)
    centers = np.array([[c * sep + j * 0.1 for j in range(d)] for c in range(nc)], dtype=np.float32)
    X = np.repeat(centers, per, axis=0) + rng.uniform(-0.5, 0.5, size=(nc * per, d)).astype(np.float32)
    return X.astype(np.float32), np.repeat(np.arange(nc), per)


@pytest.mark.parametrize("n,d,k", [(3000, 16, 10), (2000, 64, 100), (1500, 7, 33), (4000, 128, 64), (500, 3, 5),
                                   (1000, 12, 90), (800, 100, 17)])
def test_assign_brute_force_labels_identical(n, d, k):
    rng = np.random.default_rng(n * 31 + k)
    X = rng.standard_normal((n, d)).astype(np.float32)
    Cn = X[rng.choice(n, k, replace=False)] + 0.01 * rng.standard_normal((k, d)).astype(np.float32)
    Cn[1] = Cn[0]  # duplicate centroids: strict '<' keeps the lower index (kmeans.rs:364-370)
    lab_o = O.assign(X, Cn, mode="brute")
    lab_g, dist_g = vip.assign(X, Cn, mode=vip.VI_ASSIGN_REFERENCE, return_dist=True)
    assert (lab_g == lab_o).all()
    exp = np.array([O.l2sq_simd(X[i], Cn[lab_o[i]]) for i in range(0, n, 37)], dtype=np.float32)
    assert (bits(dist_g[::37]) == bits(exp)).all()


@pytest.mark.parametrize("n,d,k", [(5000, 16, 101), (3000, 64, 400), (2000, 8, 1000), (6000, 32, 257)])
def test_assign_hierarchical_matches_reference_path(n, d, k):
    """k > 100 -> assign_points_hierarchical (kmeans.rs:474-581): approximate by design, so the GPU has
    to reproduce the same hierarchy, the same top-3 meta choice and the same candidate order."""
    rng = np.random.default_rng(k)
    X = rng.standard_normal((n, d)).astype(np.float32)
    Cn = X[rng.choice(n, k, replace=False)].copy()
    for seed in (42, 7):
        lab_o = O.assign(X, Cn, seed=seed, mode="hier")
        lab_g = vip.assign(X, Cn, seed=seed, mode=vip.VI_ASSIGN_REFERENCE)
        assert (lab_g == lab_o).all(), int((lab_g != lab_o).sum())
    # the exact mode equals brute force for every k
    lab_b = O.assign(X, Cn, mode="brute")
    lab_e = vip.assign(X, Cn, mode=vip.VI_ASSIGN_EXACT)
    assert (lab_e == lab_b).all()


def test_kmeans_pp_init_identical():
    rng = np.random.default_rng(1)
    X = rng.standard_normal((3000, 24)).astype(np.float32)
    for k, seed in [(20, 42), (150, 1303)]:
        Co = O.kmeans_pp_init(X, k, seed)
        Cg, _, it = vip.kmeans_parallel(X, k, 0, seed=seed)  # max_iters = 0 returns the initial centroids
        assert it == 0 and (bits(Cg) == bits(Co)).all()
    # k > n duplicates existing centroids (kmeans.rs:216-225)
    Xs = X[:7]
    Co = O.kmeans_pp_init(Xs, 12, 5)
    Cg, _, _ = vip.kmeans_parallel(Xs, 12, 0, seed=5)
    assert (bits(Cg) == bits(Co)).all()


def test_kmeans_pp_init_sampled_variant_identical():
    """n > 50 000 switches to the sampled init (kmeans.rs:158-163, incl. its rows-0..S quirk)."""
    rng = np.random.default_rng(2)
    X = rng.standard_normal((60000, 8)).astype(np.float32)
    Co = O.kmeans_pp_init(X, 12, 42)
    Cg, _, _ = vip.kmeans_parallel(X, 12, 0, seed=42)
    assert (bits(Cg) == bits(Co)).all()


@pytest.mark.parametrize("n,d,k,iters", [(2000, 16, 20, 50), (1200, 8, 40, 30), (500, 4, 5, 300), (3000, 32, 120, 10)])
def test_mini_batch_identical(n, d, k, iters):
    rng = np.random.default_rng(n + k)
    X, _ = gaussian_clusters(rng, 5, n // 5, d, 5.0)
    rc, Co, lo, ito = O.kmeans_mini_batch(X, k, iters, seed=42)
    Cg, lg, itg = vip.kmeans_mini_batch(X, k, iters, seed=42)
    assert rc == 0 and itg == ito
    assert (bits(Cg) == bits(Co)).all()
    assert (lg == lo).all()


@pytest.mark.parametrize("n,d,k,iters", [(1500, 16, 10, 25), (600, 3, 4, 100), (2500, 24, 130, 4)])
def test_lloyd_identical(n, d, k, iters):
    rng = np.random.default_rng(n * 3 + k)
    X, _ = gaussian_clusters(rng, 4, n // 4, d, 4.0)
    rc, Co, lo, ito = O.kmeans_parallel(X, k, iters, seed=99)
    Cg, lg, itg = vip.kmeans_parallel(X, k, iters, seed=99)
    assert rc == 0 and itg == ito
    assert (bits(Cg) == bits(Co)).all()
    assert (lg == lo).all()


def _optimal(X, Cn, labels, eps=1e-5):
    # tests/test_utils/mod.rs:125-144 (sqrt-distance epsilon)
    d = np.sqrt(((X[:, None, :].astype(np.float64) - Cn[None].astype(np.float64)) ** 2).sum(-1))
    return bool((d[np.arange(len(X)), labels.astype(int)] <= d.min(axis=1) + eps).all())


def test_reference_property_tests():
    """The assertions of tests/kmeans_tests.rs that apply to any correct implementation."""
    rng = np.random.default_rng(11)
    X, truth = gaussian_clusters(rng, 3, 100, 5, 10.0)
    Cn, lab, _ = vip.kmeans_parallel(X, 3, 100, seed=42)
    assert Cn.shape == (3, 5) and lab.shape == (300,) and (lab < 3).all()          # :38-49
    assert _optimal(X, Cn, lab)
    for c in range(3):                                                              # cluster recovery :330-373
        assert len(set(lab[truth == c].tolist())) == 1
    # k = 1 -> mean of the data (:56-78)
    C1, l1, _ = vip.kmeans_parallel(X, 1, 20, seed=1)
    assert np.allclose(C1[0], X.mean(axis=0), atol=1e-3) and (l1 == 0).all()
    # k = n and k > n do not fail (:81-95,744-773)
    for k in (10, 25):
        Ck, lk, _ = vip.kmeans_parallel(X[:10], k, 5, seed=3)
        assert Ck.shape == (k, 5) and (lk < k).all()
    # identical points group together (:118-144)
    Xi = np.concatenate([np.tile(X[0], (20, 1)), np.tile(X[150], (20, 1))]).astype(np.float32)
    _, li, _ = vip.kmeans_parallel(Xi, 2, 20, seed=42)
    assert len(set(li[:20].tolist())) == 1 and len(set(li[20:].tolist())) == 1
    # mini-batch: inertia within 1.5x of full batch (:541-579)
    inertia = lambda Cc, ll: float(((X - Cc[ll.astype(int)]) ** 2).sum())
    Cm, lm, _ = vip.kmeans_mini_batch(X, 3, 100, seed=42)
    assert inertia(Cm, lm) <= 1.5 * inertia(Cn, lab) + 1e-3
    # empty input -> InvalidInput (:735-741)
    with pytest.raises(RuntimeError) as e:
        vip.kmeans_mini_batch(np.zeros((0, 4), dtype=np.float32), 3, 10)
    assert e.value.status == N.VI_ERR_INVALID_INPUT


@pytest.mark.parametrize("n,d,nlist", [(3000, 16, 0), (150, 8, 0), (12000, 32, 0), (2000, 5, 150), (1, 4, 0)])
def test_index_build_writes_the_same_files_as_the_oracle(n, d, nlist, tmp_path):
    """fit_with_paths (ivf_index.rs:58-177): same lists, same shard grouping, same bytes on disk."""
    rng = np.random.default_rng(n)
    X = rng.standard_normal((n, d)).astype(np.float32)
    ext = rng.permutation(n).astype(np.uint64) + 1000
    ts = np.where(rng.random(n) < 0.5, 0, rng.integers(1, 10 ** 9, n)).astype(np.uint64)
    o_dir, g_dir = tmp_path / "o", tmp_path / "g"
    orc = O.OracleIndex.build(X, str(o_dir / "index"), str(o_dir / "shards"), ext_ids=ext, timestamps=ts, nlist=nlist,
                              now=1_700_000_123)
    gpu = vip.build(X, str(g_dir), nlist=nlist, now_secs=1_700_000_123, ext_ids=ext, timestamps=ts)
    assert gpu.num_centroids == orc.num_centroids
    assert filecmp.cmp(o_dir / "index" / "index.bin", g_dir / "index" / "index.bin", shallow=False)
    files = sorted(os.listdir(o_dir / "shards"))
    assert files == sorted(os.listdir(g_dir / "shards")) and len(files) == orc.num_shards or n == 1
    for f in files:
        assert filecmp.cmp(o_dir / "shards" / f, g_dir / "shards" / f, shallow=False), f
    # every vector present exactly once (ivf_index_tests.rs:550-653) and searchable
    Q = X[: min(n, 64)]
    rc, Do, Io = orc.search_batch(Q, 5, 50)
    Dg, Ig = gpu.search_sync(Q, 5, 50)
    assert (Ig == Io).all() and (bits(Dg) == bits(Do)).all()
    assert (Ig[:, 0] == ext[: len(Q)].astype(np.int64)).all()


@pytest.mark.parametrize("n,d,k,kind", [(5000, 128, 300, "gauss"), (3000, 64, 130, "gauss"), (2000, 100, 256, "gauss"),
                                        (4000, 7, 500, "gauss"), (3000, 128, 1024, "sift"), (2500, 96, 200, "offset"),
                                        (2000, 32, 128, "dups"), (1500, 128, 640, "grid")])
def test_mfma_filtered_exact_assign_equals_brute_force(n, d, k, kind):
    """VI_ASSIGN_EXACT with k >= 128, d <= 128 runs the f32-MFMA filter (-2 X C^T + |c|^2) with a
    rigorous error margin and re-checks undecided rows in exact order: labels must equal
    assign_points_brute_force (kmeans.rs:462-470) on easy, near-tied and exactly tied inputs."""
    rng = np.random.default_rng(n + 7 * k)
    if kind == "gauss":
        X = rng.standard_normal((n, d)).astype(np.float32)
        Cn = X[rng.choice(n, k, replace=False)] + 0.05 * rng.standard_normal((k, d)).astype(np.float32)
    elif kind == "sift":
        X = np.clip(np.round(np.abs(rng.standard_normal((n, d)) * 40 + 20)), 0, 218).astype(np.float32)
        Cn = X[rng.choice(n, k, replace=False)].copy()
    elif kind == "offset":  # large common offset: tiny relative gaps => most rows need the exact re-check
        X = (1000.0 + rng.standard_normal((n, d))).astype(np.float32)
        Cn = (1000.0 + rng.standard_normal((k, d))).astype(np.float32)
    elif kind == "dups":    # duplicate centroids: exact ties, lowest index must win
        X = rng.standard_normal((n, d)).astype(np.float32)
        base = X[rng.choice(n, k // 2, replace=False)]
        Cn = np.concatenate([base, base])[rng.permutation(k)].copy()
    else:                   # integer grid: many exact ties between different centroids
        X = rng.integers(-2, 3, size=(n, d)).astype(np.float32)
        Cn = rng.integers(-2, 3, size=(k, d)).astype(np.float32)
    lab_o = O.assign(X, Cn, mode="brute")
    lab_g = vip.assign(X, Cn, mode=vip.VI_ASSIGN_EXACT)
    assert (lab_g == lab_o).all(), int((lab_g != lab_o).sum())


def test_exact_mode_kmeans_and_build_match_oracle_force_brute(tmp_path):
    rng = np.random.default_rng(77)
    X = rng.standard_normal((6000, 48)).astype(np.float32)
    rc, Co, lo, ito = O.kmeans_mini_batch(X, 200, 30, seed=42, force_brute=True)
    Cg, lg, itg = vip.kmeans_mini_batch(X, 200, 30, seed=42, mode=vip.VI_ASSIGN_EXACT)
    assert rc == 0 and itg == ito and (bits(Cg) == bits(Co)).all() and (lg == lo).all()


@pytest.mark.parametrize("d,kind", [(12, "gauss"), (16, "grid"), (40, "gauss"), (64, "gauss"), (72, "offset"), (100, "gauss"),
                                    (128, "grid")])
def test_candidate_sweep_exact_assign_equals_brute_force(d, kind):
    """the hi-only candidate sweep (assign_mfma.hip: first tier from 8 192 centroids on) for every chunk count 1 .. 8 —
    odd ones included, whose tile images are not a multiple of the four waves' DMA pieces — on Gaussian data, on an integer
    grid (masses of exact ties and duplicate centroids: lowest index must win, src/kmeans.rs:364-370) and far from the
    origin (wide margins: lists overflow into the older tiers): labels == assign_points_brute_force bit for bit"""
    n, k = 6000, 8192 + 77
    rng = np.random.default_rng(d)
    if kind == "grid":
        X = rng.integers(-3, 4, size=(n, d)).astype(np.float32)
        Cn = rng.integers(-3, 4, size=(k, d)).astype(np.float32)
        Cn[4000:4100] = Cn[100:200]                       # duplicate centroids
        X[:500] = Cn[rng.integers(0, k, 500)]             # points ON centroids
    else:
        X = rng.standard_normal((n, d)).astype(np.float32)
        Cn = rng.standard_normal((k, d)).astype(np.float32)
        if kind == "offset":
            X += 40.0
            Cn += 40.0
    want = O.assign(X, Cn, mode="brute")
    got = vip.assign(X, Cn, mode=vip.VI_ASSIGN_EXACT)
    bad = np.nonzero(got != want)[0]
    assert bad.size == 0, f"{bad.size} of {n} labels differ, first row {bad[0]}: {got[bad[0]]} vs {want[bad[0]]}"
