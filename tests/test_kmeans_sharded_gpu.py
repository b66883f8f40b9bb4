"""k-means on device-resident and on SHARDED points through the C ABI on one GPU:

  * vi_kmeans_{mini_batch,parallel}_device == the host-pointer entries, bit for bit;
  * N in-process ranks (one thread per rank, every rank its own slice of the points in its own device buffers; the
    collectives emulated by a thread barrier + host staging, summed in rank order as a ring all-reduce would) driven by
    vector_indexer_py.distributed.kmeans_*_sharded:
      - mini-batch: centroids AND labels identical to the single-GPU run (the training loop reads the same rows);
      - Lloyd: world 1 identical; worlds 2 / 8: first-iteration labels identical, centroids equal to rounding
        (rtol 1e-5: the all-reduce associates the per-cluster sums differently), same inertia to 1e-3.
"""
import ctypes as C
import threading

import numpy as np
import pytest

import oracle_lib as O
import vector_indexer_py as vip
from hiprt import Hip
from vector_indexer_py import _native as N
from vector_indexer_py import distributed as VD

pytestmark = pytest.mark.gpu


def clustered(n, d, nc, seed):
    rng = np.random.default_rng(seed)
    centers = rng.standard_normal((nc, d)).astype(np.float32) * 4
    return (centers[rng.integers(0, nc, n)] + rng.standard_normal((n, d)).astype(np.float32)).astype(np.float32)


@pytest.mark.parametrize("n,d,k,mode", [(6000, 32, 40, 0), (5000, 64, 150, 0), (4000, 128, 200, 1), (300, 7, 12, 0)])
def test_device_entries_equal_host_entries(n, d, k, mode):
    X = clustered(n, d, 25, n + k)
    hip = Hip()
    try:
        Xd = hip.upload(X)
        Cd, Ld = hip.alloc(k * d * 4), hip.alloc(n * 4)
        for name, host_fn in (("vi_kmeans_mini_batch_device", vip.kmeans_mini_batch),
                              ("vi_kmeans_parallel_device", vip.kmeans_parallel)):
            Ch, lh, ith = host_fn(X, k, 12, seed=7, mode=mode)
            it = C.c_uint64(0)
            N.check(getattr(N.lib(), name)(0, Xd, n, d, k, 12, -1.0, 7, mode, Cd, Ld, C.byref(it)))
            assert it.value == ith
            assert (hip.download(Cd, (k, d), np.float32).view(np.uint32) == Ch.view(np.uint32)).all(), name
            assert (hip.download(Ld, (n,), np.uint32) == lh.astype(np.uint32)).all(), name
    finally:
        hip.close()


class ThreadComm:
    """`world` in-process ranks on one GPU: collectives = barrier + host staging, reductions in rank order"""

    class Shared:
        def __init__(self, world, hip):
            self.world, self.hip = world, hip
            self.barrier = threading.Barrier(world)
            self.slots = [None] * world
            self.lock = threading.Lock()

    class Buf:
        def __init__(self, ptr):
            self.ptr = ptr

    def __init__(self, shared, rank):
        self.s, self.rank, self.world = shared, rank, shared.world

    def alloc(self, nbytes):
        with self.s.lock:
            return ThreadComm.Buf(self.s.hip.alloc(max(nbytes, 8)))

    def _exchange(self, mine):
        self.s.slots[self.rank] = mine
        self.s.barrier.wait()
        parts = list(self.s.slots)
        self.s.barrier.wait()
        return parts

    def all_reduce_sum(self, buf, count, kind):
        dt = np.float32 if kind == "f32" else np.uint32
        parts = self._exchange(self.s.hip.download(buf.ptr, (count,), dt))
        acc = parts[0].copy()
        for p in parts[1:]:
            acc += p
        self.s.hip.upload_to(buf.ptr, acc)

    def broadcast(self, buf, nbytes, root=0):
        parts = self._exchange(self.s.hip.download(buf.ptr, (nbytes,), np.uint8) if self.rank == root else None)
        self.s.hip.upload_to(buf.ptr, parts[root])

    def copy(self, dst, src, nbytes):
        assert self.s.hip.rt.hipMemcpy(dst, src, nbytes, 3) == 0

    def fetch_rows(self, pts, rows, out_ptr):
        rows = rows.astype(np.int64)
        mine = (rows >= pts.row_begin) & (rows < pts.row_begin + pts.n_local)
        buf = np.zeros((rows.size, pts.d), dtype=np.int32)
        buf[mine] = pts.tensor.view(np.int32)[rows[mine] - pts.row_begin]   # pts.tensor: the slice on the host
        parts = self._exchange(buf)
        acc = parts[0].copy()
        for p in parts[1:]:
            acc += p
        self.s.hip.upload_to(out_ptr, acc)


def run_ranks(world, X, fn):
    """fn(engine, comm, pts) on `world` threads over contiguous slices of X -> list of per-rank results"""
    hip = Hip()
    shared = ThreadComm.Shared(world, hip)
    n, d = X.shape
    per = (n + world - 1) // world
    out, errs = [None] * world, []

    def body(r):
        try:
            b, e = min(n, r * per), min(n, (r + 1) * per)
            Xl = np.ascontiguousarray(X[b:e])
            with shared.lock:
                ptr = hip.upload(Xl) if e > b else hip.alloc(8)
            pts = VD.ShardedPoints(ptr, e - b, d, b, n, tensor=Xl)
            out[r] = fn(VD.GpuKMeansEngine(0), ThreadComm(shared, r), pts, hip)
        except BaseException as ex:  # noqa: BLE001
            errs.append(ex)
            shared.barrier.abort()
    ts = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    if errs:
        hip.close()
        raise errs[0]
    return out, hip


@pytest.mark.parametrize("world", [2, 8])
@pytest.mark.parametrize("n,d,k,mode", [(9000, 32, 60, 0), (70000, 16, 130, 1)])
def test_sharded_mini_batch_equals_single_gpu(world, n, d, k, mode):
    """n = 70 000 takes the sampled k-means++ (kmeans.rs:158-163): its candidate rows live on every rank"""
    X = clustered(n, d, 30, 3 * n + world)
    Cs, ls, its = vip.kmeans_mini_batch(X, k, 15, seed=42, mode=mode)

    def fn(engine, comm, pts, hip):
        Cb, Lb, it = VD.kmeans_mini_batch_sharded(engine, comm, pts, k, 15, seed=42, mode=mode)
        return hip.download(Cb.ptr, (k, d), np.float32), hip.download(Lb.ptr, (max(pts.n_local, 1),), np.uint32)[:pts.n_local], it
    res, hip = run_ranks(world, X, fn)
    try:
        for Cr, _, it in res:
            assert it == its and (Cr.view(np.uint32) == Cs.view(np.uint32)).all()
        assert (np.concatenate([r[1] for r in res]) == ls.astype(np.uint32)).all()
    finally:
        hip.close()


@pytest.mark.parametrize("world", [1, 2, 8])
def test_sharded_lloyd_all_reduce(world):
    n, d, k = 8000, 24, 50
    X = clustered(n, d, 20, 11)
    X[:40] = 1000.0 + np.arange(40, dtype=np.float32)[:, None]  # far outliers: k-means++ picks them, some clusters run empty later
    for max_iters in (1, 25):
        Cs, ls, its = vip.kmeans_parallel(X, k, max_iters, seed=5, mode=vip.VI_ASSIGN_EXACT)

        def fn(engine, comm, pts, hip):
            Cb, Lb, it = VD.kmeans_parallel_sharded(engine, comm, pts, k, max_iters, seed=5, mode=vip.VI_ASSIGN_EXACT)
            return hip.download(Cb.ptr, (k, d), np.float32), hip.download(Lb.ptr, (max(pts.n_local, 1),), np.uint32)[:pts.n_local], it
        res, hip = run_ranks(world, X, fn)
        try:
            lab = np.concatenate([r[1] for r in res])
            for Cr, _, it in res:
                assert (Cr.view(np.uint32) == res[0][0].view(np.uint32)).all()   # every rank holds the same table
            Cr, it = res[0][0], res[0][2]
            if world == 1:
                assert it == its and (Cr.view(np.uint32) == Cs.view(np.uint32)).all() and (lab == ls).all()
            elif max_iters == 1:
                assert (lab == ls).all()                     # same initial centroids => same first assignment
                assert np.allclose(Cr, Cs, rtol=1e-5, atol=1e-6)
            else:
                def inertia(Cn, l):
                    return float(((X - Cn[l.astype(np.int64)]) ** 2).sum())
                assert abs(inertia(Cr, lab) / inertia(Cs, ls) - 1.0) < 1e-3
        finally:
            hip.close()


def test_sharded_lloyd_with_ranks_that_own_no_points():
    """n = 9 points over 8 ranks (two per rank: ranks 5..7 are empty): an empty rank contributes zero sums and counts to
    the all-reduce instead of failing the loop (vi_kmeans_partial_sums_device with n_local = 0)"""
    n, d, k, world = 9, 8, 3, 8
    X = clustered(n, d, 3, 5)
    Cs, ls, its = vip.kmeans_parallel(X, k, 6, seed=9, mode=vip.VI_ASSIGN_EXACT)

    def fn(engine, comm, pts, hip):
        Cb, Lb, it = VD.kmeans_parallel_sharded(engine, comm, pts, k, 6, seed=9, mode=vip.VI_ASSIGN_EXACT)
        return hip.download(Cb.ptr, (k, d), np.float32), hip.download(Lb.ptr, (max(pts.n_local, 1),), np.uint32)[:pts.n_local], it
    res, hip = run_ranks(world, X, fn)
    try:
        assert [r[1].size for r in res] == [2, 2, 2, 2, 1, 0, 0, 0]
        for Cr, _, it in res:
            assert (Cr.view(np.uint32) == res[0][0].view(np.uint32)).all()
        lab = np.concatenate([r[1] for r in res])

        def inertia(Cn, l):
            return float(((X - Cn[l.astype(np.int64)]) ** 2).sum())
        assert inertia(res[0][0], lab) <= inertia(Cs, ls) * (1 + 1e-3) + 1e-6
    finally:
        hip.close()


def test_rng_stream_is_the_oracles():
    """vi_rng (rand 0.8.5 StdRng restated in rng.hpp) draws what the oracle's restatement draws (itself pinned to the
    golden vectors of tests/golden/rng.json by test_oracle_golden.py)"""
    for seed in (0, 42, 2 ** 40 + 7):
        h = N.lib().vi_rng_seed_from_u64(seed)
        st = O.OrcRng()
        O.lib().orc_rng_seed_from_u64(C.byref(st), seed)
        for hi in (1, 2, 10, 1000, 2 ** 20 + 3, 2 ** 33, 50_000):
            assert N.lib().vi_rng_gen_range(h, 0, hi) == O.lib().orc_rng_gen_range_usize(C.byref(st), 0, hi)
        N.lib().vi_rng_free(h)
