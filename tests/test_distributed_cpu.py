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
