#!/usr/bin/env python3
"""Run a handful of searches on the bench workload (for rocprofv3 passes)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "vector-indexer_amd")]
import bench  # noqa: E402
import vector_indexer_py as vip  # noqa: E402

dev = torch.device("cuda", 0)
n, d, nlist, nq, k = 1_000_000, 128, 4096, 10000, 10
n_probe = int(os.environ.get("NPROBE", 16))
xb, xq = bench.make_dataset(n, d, nq, 42, dev)
work = "/tmp/vi_scan_bench"
if not os.path.exists(work + "/index/index.bin"):
    vip.build(xb.cpu().numpy(), work, nlist=nlist, now_secs=1_700_000_000)
index = vip.load(work + "/index", work + "/shards", d)
D = torch.empty((nq, k), dtype=torch.float32, device=dev)
I = torch.empty((nq, k), dtype=torch.int64, device=dev)
for _ in range(4):
    index.search_device(xq.data_ptr(), nq, k, n_probe, D.data_ptr(), I.data_ptr(), 0)
print("done")
