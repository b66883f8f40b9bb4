"""CPU (gloo, world_size 2): the multi-GPU search protocol — every list striped over the ranks, all-gather of per-rank
top-k, merge on (dist, tie) — reproduces the single-index result.  The per-rank searcher is the
oracle's partial search (the GPU kernels are covered by the -m gpu tests)."""
import os
import socket
import sys

import numpy as np
import pytest


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, work, out_path):
    import torch
    import torch.distributed as dist
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [os.path.join(root, "tests"), os.path.join(root, "vector-indexer_amd")]
    import oracle_lib as O
    from vector_indexer_py.distributed import ShardedSearcher, merge_partials_reference, block_owner

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    orc = O.OracleIndex.load(os.path.join(work, "index"), os.path.join(work, "shards"))
    _, c2s = orc.centroids()
    assert [block_owner(b, world) for b in range(5)] == [b % world for b in range(5)]
    Q = np.load(os.path.join(work, "q.npy"))

    def local_search(xq, k, n_probe):
        D, I, T = orc.search_partial_batch(xq.numpy(), k, n_probe, rank, world)
        return torch.from_numpy(D), torch.from_numpy(I), torch.from_numpy(T.view(np.int64))

    def merge(Dg, Ig, Tg):
        D, I = merge_partials_reference(Dg.numpy(), Ig.numpy(), Tg.numpy())
        return torch.from_numpy(D), torch.from_numpy(I)

    s = ShardedSearcher(local_search, merge)
    res = {}
    for k, p in [(10, 8), (5, 40), (20, 3)]:
        D, I = s.search(torch.from_numpy(Q), k, p)
        res[f"D_{k}_{p}"], res[f"I_{k}_{p}"] = D.numpy(), I.numpy()
    if rank == 0:
        np.savez(out_path, **res)
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_search_protocol_gloo_world2(tmp_path):
    import torch.multiprocessing as mp
    import oracle_lib as O
    rng = np.random.default_rng(4)
    base = rng.integers(-4, 5, size=(1500, 12)).astype(np.float32)       # integer grid: many exact ties
    X = np.concatenate([base, base[:300]])
    work = str(tmp_path)
    orc = O.OracleIndex.build(X, os.path.join(work, "index"), os.path.join(work, "shards"), nlist=36)
    assert orc.num_shards >= 2
    Q = np.concatenate([base[:40], rng.integers(-4, 5, size=(24, 12)).astype(np.float32)])
    np.save(os.path.join(work, "q.npy"), Q)
    out = os.path.join(work, "out.npz")
    mp.spawn(_worker, args=(2, _free_port(), work, out), nprocs=2, join=True)
    got = np.load(out)
    for k, p in [(10, 8), (5, 40), (20, 3)]:
        rc, D, I = orc.search_batch(Q, k, p)
        assert rc == 0
        assert (got[f"I_{k}_{p}"] == I).all(), (k, p)
        assert (got[f"D_{k}_{p}"].view(np.uint32) == D.view(np.uint32)).all()


# ---------------------------------------------------------------------------------------------------------------
# sharded k-means: the exchange of vector_indexer_py.distributed.kmeans_parallel_sharded over gloo,
# world 2, with the oracle as the per-rank arithmetic (the C-ABI engine is covered by tests/test_kmeans_sharded_gpu.py)
# ---------------------------------------------------------------------------------------------------------------
class _OracleEngine:
    """GpuKMeansEngine's interface on host memory: oracle arithmetic, numpy reductions"""

    def __init__(self, O):
        import ctypes as C
        self.O, self.C = O, C

    def _f32(self, ptr, n):
        return np.ctypeslib.as_array((self.C.c_float * n).from_address(ptr))

    def _u32(self, ptr, n):
        return np.ctypeslib.as_array((self.C.c_uint32 * n).from_address(ptr))

    def pp_init(self, fetch, n_global, d, k, seed, C_ptr):
        buf = np.zeros((n_global, d), dtype=np.float32)         # (a CPU engine may fetch every row: the data is tiny)
        fetch(np.arange(n_global, dtype=np.uint64), buf.ctypes.data)
        self._f32(C_ptr, k * d)[:] = self.O.kmeans_pp_init(buf, k, seed).reshape(-1)

    def assign(self, X_ptr, n_local, d, C_ptr, k, seed, mode, labels_ptr):
        if n_local:
            lab = self.O.assign(self._f32(X_ptr, n_local * d).reshape(n_local, d), self._f32(C_ptr, k * d).reshape(k, d),
                                mode="brute")
            self._u32(labels_ptr, n_local)[:] = lab.astype(np.uint32)

    def partial_sums(self, X_ptr, n_local, d, labels_ptr, k, sums_ptr, counts_ptr):
        S, Nc = self._f32(sums_ptr, k * d).reshape(k, d), self._u32(counts_ptr, k)
        S[:] = 0
        Nc[:] = 0
        if n_local:
            X, lab = self._f32(X_ptr, n_local * d).reshape(n_local, d), self._u32(labels_ptr, n_local)
            for i in range(n_local):                             # ascending i inside the rank (kmeans.rs:693-697)
                S[lab[i]] += X[i]
                Nc[lab[i]] += 1

    def finish_update(self, sums_ptr, counts_ptr, k, d, C_prev_ptr, C_new_ptr):
        S, Nc = self._f32(sums_ptr, k * d).reshape(k, d), self._u32(counts_ptr, k)
        Cn = self._f32(C_new_ptr, k * d).reshape(k, d)
        Cn[:] = 0
        nz = Nc > 0
        Cn[nz] = S[nz] / Nc[nz].astype(np.float32)[:, None]
        return np.nonzero(~nz)[0].astype(np.uint32)

    def centroid_delta(self, C_new_ptr, C_prev_ptr, k, d):
        a, b = self._f32(C_new_ptr, k * d).reshape(k, d), self._f32(C_prev_ptr, k * d).reshape(k, d)
        dsq = np.float32(0)
        for c in range(k):
            acc = np.float32(0)
            for j in range(d):
                t = np.float32(a[c, j] - b[c, j])
                acc = np.float32(acc + t * t)
            dsq = np.float32(dsq + acc)
        return float(np.sqrt(np.float32(dsq / np.float32(k * d))))

    def rng(self, seed):
        O, C = self.O, self.C
        st = O.OrcRng()
        O.lib().orc_rng_seed_from_u64(C.byref(st), seed)

        class R:
            def gen_range(s, lo, hi):
                return int(O.lib().orc_rng_gen_range_usize(C.byref(st), lo, hi))
        return R()


def _kmeans_worker(rank, world, port, work):
    import torch
    import torch.distributed as dist
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [os.path.join(root, "tests"), os.path.join(root, "vector-indexer_amd")]
    import oracle_lib as O
    from vector_indexer_py import distributed as VD

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    X = np.load(os.path.join(work, "x.npy"))
    n, d = X.shape
    per = (n + world - 1) // world
    b, e = min(n, rank * per), min(n, (rank + 1) * per)
    Xl = torch.from_numpy(np.ascontiguousarray(X[b:e]))
    pts = VD.ShardedPoints(Xl.data_ptr(), e - b, d, b, n, tensor=Xl)
    comm = VD.TorchComm("cpu")
    # the row exchange alone: rows owned by either rank, repeated and out of order
    rows = np.array([0, n - 1, per, per - 1, 5, 5, n // 3], dtype=np.uint64)
    got = torch.zeros((rows.size, d), dtype=torch.float32)
    comm.fetch_rows(pts, rows, got.data_ptr())
    assert (got.numpy().view(np.uint32) == X[rows.astype(np.int64)].view(np.uint32)).all()
    res = {}
    for iters in (1, 12):
        Cb, Lb, it = VD.kmeans_parallel_sharded(_OracleEngine(O), comm, pts, 9, iters, seed=3)
        res[f"C{iters}"] = Cb.t[:9 * d].view(torch.float32).numpy().reshape(9, d).copy()
        res[f"L{iters}"] = Lb.t[:e - b].numpy().astype(np.uint32).copy()
        res[f"it{iters}"] = np.array([it])
    np.savez(os.path.join(work, f"km_{rank}.npz"), **res)
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_lloyd_exchange_gloo_world2(tmp_path):
    import torch.multiprocessing as mp
    import oracle_lib as O
    rng = np.random.default_rng(8)
    centers = rng.standard_normal((9, 6)).astype(np.float32) * 5
    X = (centers[rng.integers(0, 9, 700)] + rng.standard_normal((700, 6)).astype(np.float32)).astype(np.float32)
    X[:3] = 500.0                                               # duplicates far away: a cluster that runs empty
    work = str(tmp_path)
    np.save(os.path.join(work, "x.npy"), X)
    mp.spawn(_kmeans_worker, args=(2, _free_port(), work), nprocs=2, join=True)
    r0, r1 = np.load(os.path.join(work, "km_0.npz")), np.load(os.path.join(work, "km_1.npz"))
    for iters in (1, 12):
        rc, Co, lo, ito = O.kmeans_parallel(X, 9, iters, seed=3, force_brute=True)
        assert rc == 0
        assert (r0[f"C{iters}"].view(np.uint32) == r1[f"C{iters}"].view(np.uint32)).all()   # ranks agree bit for bit
        lab = np.concatenate([r0[f"L{iters}"], r1[f"L{iters}"]])
        if iters == 1:
            assert (lab == lo).all()
            assert np.allclose(r0["C1"], Co, rtol=1e-5, atol=1e-6)
        else:
            def inertia(Cn, l):
                return float(((X - Cn[l.astype(np.int64)]) ** 2).sum())
            assert abs(inertia(r0["C12"], lab) / inertia(Co, lo) - 1.0) < 1e-3


def _kmeans_tiny_worker(rank, world, port, work):
    import torch
    import torch.distributed as dist
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [os.path.join(root, "tests"), os.path.join(root, "vector-indexer_amd")]
    import oracle_lib as O
    from vector_indexer_py import distributed as VD

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    X = np.load(os.path.join(work, "x.npy"))
    n, d = X.shape
    per = (n + world - 1) // world
    b, e = min(n, rank * per), min(n, (rank + 1) * per)
    Xl = torch.from_numpy(np.ascontiguousarray(X[b:e]) if e > b else np.zeros((1, d), np.float32))
    pts = VD.ShardedPoints(Xl.data_ptr(), e - b, d, b, n, tensor=Xl)
    Cb, Lb, it = VD.kmeans_parallel_sharded(_OracleEngine(O), VD.TorchComm("cpu"), pts, 2, 4, seed=3)
    np.savez(os.path.join(work, f"tiny_{rank}.npz"), C=Cb.t[:2 * d].view(torch.float32).numpy().reshape(2, d).copy(),
             L=Lb.t[:e - b].numpy().astype(np.uint32).copy(), n_local=np.array([e - b]))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_lloyd_gloo_world_larger_than_n(tmp_path):
    """world 3 over n = 2 points (one per rank: rank 2 owns nothing): the empty rank contributes zero sums and counts to the
    all-reduce (run_kmeans_parallel's update, src/kmeans.rs:674-719, summed over ranks) instead of failing the loop"""
    import torch.multiprocessing as mp
    X = np.array([[0.0, 0.0, 1.0], [10.0, 10.0, 10.0]], dtype=np.float32)
    work = str(tmp_path)
    np.save(os.path.join(work, "x.npy"), X)
    mp.spawn(_kmeans_tiny_worker, args=(3, _free_port(), work), nprocs=3, join=True)
    r = [np.load(os.path.join(work, f"tiny_{i}.npz")) for i in range(3)]
    assert [int(x["n_local"][0]) for x in r] == [1, 1, 0]
    assert (r[0]["C"].view(np.uint32) == r[1]["C"].view(np.uint32)).all() and (r[0]["C"].view(np.uint32) == r[2]["C"].view(np.uint32)).all()
    lab = np.concatenate([x["L"] for x in r])
    assert sorted(lab.tolist()) == [0, 1]                       # two points, two clusters: each its own
    assert np.allclose(np.sort(r[0]["C"], axis=0), np.sort(X, axis=0))
