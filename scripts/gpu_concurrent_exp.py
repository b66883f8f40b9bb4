#!/usr/bin/env python3
"""Diagnostic: throughput of T host threads issuing batches on ONE index handle (the library holds 4 search contexts, each
with its own stream), against one thread.  Usage: gpu_concurrent_exp.py [nprobe]"""
import os
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "vector-indexer_amd")]
import bench  # noqa: E402
import vector_indexer_py as vip  # noqa: E402

dev = torch.device("cuda", 0)
n, d, nlist, nq, k = 1_000_000, 128, 4096, 10000, 10
P = int(sys.argv[1]) if len(sys.argv) > 1 else 32
xb, xq = bench.make_dataset(n, d, nq, 42, dev)
work = f"/tmp/vi_rank_exp_{n}_{d}_{nlist}_0"
if not os.path.exists(work + "/index/index.bin"):
    vip.build(xb.cpu().numpy(), work, nlist=nlist, now_secs=1_700_000_000)
index = vip.load(work + "/index", work + "/shards", d)
for T in (1, 2, 3, 4):
    outs = [(torch.empty((nq, k), dtype=torch.float32, device=dev), torch.empty((nq, k), dtype=torch.int64, device=dev)) for _ in range(T)]
    steps = 40

    def worker(t):
        D, I = outs[t]
        for _ in range(steps):
            index.search_device(xq.data_ptr(), nq, k, P, D.data_ptr(), I.data_ptr(), 0)

    for t in range(T):
        worker.__call__(t) if False else None
    worker(0)  # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    th = [threading.Thread(target=worker, args=(t,)) for t in range(T)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print(f"threads {T}: {T * steps * nq / el / 1e6:.2f} M queries/s, {el / (T * steps) * 1e3:.3f} ms per batch", flush=True)
    same = all(torch.equal(outs[0][1], o[1]) for o in outs)
    print("   identical results across threads:", same)
