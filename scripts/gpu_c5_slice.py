#!/usr/bin/env python3
"""One rank's share of BASELINE config C5 on one MI355X: N = 1e8 / 8 = 1.25e7 vectors, D = 96, nlist = 65536, X ~ N(0,1)
(seed 42), built and searched through the product (vip.build -> GPU k-means, GPU list build, shard export; search of
10 000 queries at nprobe 32, k 10).  Prints one JSON object: build phases, file bytes, search phase times, recall against
an exact brute force.  (The 8-GPU run itself needs a node the builder cannot launch; this is the per-rank slice of it.)

    python scripts/gpu_c5_slice.py [--n 12500000] [--nlist 65536] [--exact-assign]
"""
import argparse
import json
import os
import shutil
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "vector-indexer_amd")]
import bench  # noqa: E402
import vector_indexer_py as vip  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=12_500_000)
ap.add_argument("--d", type=int, default=96)
ap.add_argument("--nlist", type=int, default=65536)
ap.add_argument("--nq", type=int, default=10_000)
ap.add_argument("--k", type=int, default=10)
ap.add_argument("--nprobe", type=int, default=32)
ap.add_argument("--exact-assign", action="store_true", help="VI_ASSIGN_EXACT for the final assignment (the reference uses its 2-level approximation above 100 lists)")
ap.add_argument("--work-dir", default="/tmp/vi_c5_slice")
a = ap.parse_args()

dev = torch.device("cuda", 0)
g = torch.Generator(device=dev)
g.manual_seed(42)
xb = torch.randn(a.n, a.d, generator=g, device=dev)
xq = torch.randn(a.nq, a.d, generator=g, device=dev)
shutil.rmtree(a.work_dir, ignore_errors=True)
xb_h = xb.cpu().numpy()
t0 = time.time()
idx = vip.build(xb_h, a.work_dir, nlist=a.nlist, now_secs=1_700_000_000, assign_mode=1 if a.exact_assign else 0)
build_s = time.time() - t0
bs = idx.build_stats()
del xb_h
files = os.listdir(os.path.join(a.work_dir, "shards"))
file_bytes = sum(os.path.getsize(os.path.join(a.work_dir, "shards", f)) for f in files)
idx.enable_timing(True)
D = torch.empty((a.nq, a.k), dtype=torch.float32, device=dev)
I = torch.empty((a.nq, a.k), dtype=torch.int64, device=dev)
res = {}
for p in sorted({8, 16, a.nprobe, 64}):
    acc = []
    for r in range(6):
        idx.search_device(xq.data_ptr(), a.nq, a.k, p, D.data_ptr(), I.data_ptr(), 0)
        st = idx.last_stats()
        if r >= 2:
            acc.append([st[f] for f in ("ms_total", "ms_coarse", "ms_group", "ms_scan", "ms_merge")])
    m = np.mean(np.array(acc), axis=0)
    res[p] = {"ms": {"total": round(float(m[0]), 3), "coarse": round(float(m[1]), 3), "grouping": round(float(m[2]), 3),
                     "list_rank": round(float(m[3]), 3), "select": round(float(m[4]), 3)},
              "queries_per_s": round(a.nq / (float(m[0]) * 1e-3), 1), "scanned_vectors_per_query": round(st["scanned_vectors"] / a.nq, 1),
              "rank_mode": int(st["rank_mode"]), "queries_per_work_item": int(st["group_queries"]), "I": I.clone()}
gt = bench.ground_truth(xb, xq, a.k)
for p in res:
    r1, ri = bench.recalls(res[p].pop("I"), gt)
    res[p]["recall_1nn_at_k"], res[p]["recall_at_k"] = round(r1, 4), round(ri, 4)
print(json.dumps({"workload": f"C5 per-rank slice: N={a.n} D={a.d} nlist={a.nlist} k={a.k} nq={a.nq}, X~N(0,1) seed 42, "
                              f"assign_mode={'exact' if a.exact_assign else 'reference (2-level)'}",
                  "build_s": round(build_s, 2), "build_phases_ms": {k_[3:]: round(v, 1) for k_, v in bs.items() if k_.startswith("ms_")},
                  "lists": int(bs["lists"]), "shards": int(bs["shards"]), "shard_files": len(files), "shard_file_bytes": file_bytes,
                  "resident_vectors": idx.num_vectors, "search": res}), flush=True)
shutil.rmtree(a.work_dir, ignore_errors=True)
