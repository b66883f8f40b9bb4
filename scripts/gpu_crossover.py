#!/usr/bin/env python3
"""Diagnostic: batch size at which the MFMA engine overtakes the exact-order VALU engine."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "vector-indexer_amd")]
import bench  # noqa: E402
import vector_indexer_py as vip  # noqa: E402

dev = torch.device("cuda", 0)
n, d, nlist, k = 1_000_000, 128, 4096, 10
xb, xq = bench.make_dataset(n, d, 10000, 42, dev)
work = "/tmp/vi_scan_bench"
if not os.path.exists(work + "/index/index.bin"):
    vip.build(xb.cpu().numpy(), work, nlist=nlist, now_secs=1_700_000_000)
index = vip.load(work + "/index", work + "/shards", d)
for nq in (1, 8, 32, 64, 128, 256, 512, 1024, 2048, 4096):
    D = torch.empty((nq, k), dtype=torch.float32, device=dev)
    I = torch.empty((nq, k), dtype=torch.int64, device=dev)
    row = []
    for eng in ("0", "1"):
        os.environ["VI_FILTER"] = eng
        for _ in range(3):
            index.search_device(xq.data_ptr(), nq, k, 16, D.data_ptr(), I.data_ptr(), 0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 20
        for _ in range(reps):
            index.search_device(xq.data_ptr(), nq, k, 16, D.data_ptr(), I.data_ptr(), 0)
        torch.cuda.synchronize()
        row.append((time.perf_counter() - t0) / reps * 1e3)
    print(f"nq={nq:5d} pairs/list={nq * 16 / 4094:6.2f}  VALU {row[0]:7.3f} ms   MFMA {row[1]:7.3f} ms", flush=True)
