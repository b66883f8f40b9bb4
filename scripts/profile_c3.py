#!/usr/bin/env python3
"""BASELINE config C3 alone, for rocprofv3 (program directly after `--`): N=1e7 D=128 k=16384, X ~ N(0,1) seed 42 on the
device; W warm-up + K timed passes of (1) the exact nearest-centroid assign (vi_assign_device, VI_ASSIGN_EXACT: bf16x3 MFMA
tier -> f32 MFMA tier -> exact scan) and (2) the update pass (vi_kmeans_partial_sums_device: grouping + ascending-id sums).
Nothing else touches the GPU except torch's generator kernels.  Prints one JSON object.

    python3 scripts/profile_c3.py [--n 10000000] [--passes 3] [--warmup 1] [--train]
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vector-indexer_amd")]
from vector_indexer_py import _native  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=10_000_000)
ap.add_argument("--d", type=int, default=128)
ap.add_argument("--k", type=int, default=16384)
ap.add_argument("--passes", type=int, default=3)
ap.add_argument("--warmup", type=int, default=1)
ap.add_argument("--train", action="store_true", help="also one vi_kmeans_mini_batch_device run (20 iterations + final assign)")
a = ap.parse_args()
lib = _native.lib()
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev)
g.manual_seed(42)
X = torch.randn(a.n, a.d, generator=g, device=dev)
Cn = X[torch.randperm(a.n, generator=g, device=dev)[:a.k]].contiguous()
lab = torch.empty(a.n, dtype=torch.int32, device=dev)
sums = torch.empty((a.k, a.d), dtype=torch.float32, device=dev)
cnts = torch.empty(a.k, dtype=torch.int32, device=dev)
torch.cuda.synchronize()
st = _native.AssignStats()
assign, update = [], []
for i in range(a.warmup + a.passes):
    _native.check(lib.vi_assign_device(0, X.data_ptr(), a.n, a.d, Cn.data_ptr(), a.k, 42, 1, lab.data_ptr(), C.byref(st)))
    t0 = time.perf_counter()
    _native.check(lib.vi_kmeans_partial_sums_device(0, X.data_ptr(), a.n, a.d, lab.data_ptr(), a.k, sums.data_ptr(), cnts.data_ptr()))
    t1 = time.perf_counter()
    if i >= a.warmup:
        assign.append((st.ms_total, st.ms_filter, int(st.tier1_rows), int(st.ambiguous_rows)))
        update.append((t1 - t0) * 1e3)
flops = 2.0 * a.n * a.k * a.d
ub = 4.0 * a.n * a.d + 8.0 * a.n + 4.0 * a.k * a.d
best = min(assign)
out = {"workload": f"C3 N={a.n} D={a.d} k={a.k}", "passes": a.passes,
       "assign_ms_total": [round(x[0], 2) for x in assign], "assign_ms_first_tier": [round(x[1], 2) for x in assign],
       "rows_left_by_first_tier": best[2], "rows_re_evaluated_exactly": best[3],
       "first_tier_useful_TFLOPs": round(flops / (min(x[1] for x in assign) * 1e-3) / 1e12, 1),
       "update_ms": [round(x, 2) for x in update], "update_algorithmic_GBps": round(ub / (min(update) * 1e-3) / 1e9, 1),
       "counts_sum": int(cnts.sum().item()), "largest_clusters": sorted(cnts.tolist())[-8:],
       "clusters_over_4096": int((cnts > 4096).sum().item()), "points_in_them": int(cnts[cnts > 4096].sum().item())}
if a.train:
    Cout = torch.empty((a.k, a.d), dtype=torch.float32, device=dev)
    it = C.c_uint64(0)
    t0 = time.perf_counter()
    _native.check(lib.vi_kmeans_mini_batch_device(0, X.data_ptr(), a.n, a.d, a.k, 20, -1.0, 42, 0, Cout.data_ptr(), lab.data_ptr(), C.byref(it)))
    out["mini_batch_train_s"] = round(time.perf_counter() - t0, 2)
print(json.dumps(out), flush=True)
