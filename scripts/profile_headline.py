#!/usr/bin/env python3
"""The headline operating point alone, for rocprofv3 (program directly after `--`): BASELINE config C2 (N=1e6 D=128
nlist=4096, SIFT-shaped synthetic data of bench.py), W warm-up + K timed steps of 10 000 device-resident queries at
nprobe 32, k 10 — no nprobe sweep, no ground truth, no CPU baseline.  Every launch of the search kernels in the trace
is therefore a launch at the headline operating point (the index build's k-means kernels have other names).
Prints one JSON object with the HIP-event phase times the library reports (means over the timed steps).

    python3 scripts/profile_headline.py [--steps 10] [--warmup 2] [--nprobe 32] [--real-valued]
"""
import argparse
import json
import os
import shutil
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vector-indexer_amd")]
import bench  # noqa: E402
import vector_indexer_py as vip  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1_000_000)
ap.add_argument("--d", type=int, default=128)
ap.add_argument("--nlist", type=int, default=4096)
ap.add_argument("--nq", type=int, default=10_000)
ap.add_argument("--k", type=int, default=10)
ap.add_argument("--nprobe", type=int, default=32)
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--warmup", type=int, default=2)
ap.add_argument("--real-valued", action="store_true", help="VI_FILTER_HI_ONLY=0: the bf16 x 3 ranking real-valued lists take")
ap.add_argument("--no-timing", action="store_true", help="no phase events on the search stream (the phase times read 0)")
a = ap.parse_args()
if a.real_valued:
    os.environ["VI_FILTER_HI_ONLY"] = "0"
dev = torch.device("cuda", 0)
xb, xq = bench.make_dataset(a.n, a.d, a.nq, 42, dev)
work = os.path.join(tempfile.gettempdir(), f"vi_prof_{os.getuid()}_{a.n}_{a.d}_{a.nlist}")
shutil.rmtree(work, ignore_errors=True)
idx = vip.build(xb.cpu().numpy(), work, nlist=a.nlist, now_secs=1_700_000_000)
del xb
idx.enable_timing(not a.no_timing)
D = torch.empty((a.nq, a.k), dtype=torch.float32, device=dev)
I = torch.empty((a.nq, a.k), dtype=torch.int64, device=dev)
acc = []
torch.cuda.synchronize()
for s in range(a.warmup + a.steps):
    if s == a.warmup:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
    idx.search_device(xq.data_ptr(), a.nq, a.k, a.nprobe, D.data_ptr(), I.data_ptr(), 0)
    st = idx.last_stats()
    if s >= a.warmup:
        acc.append([st[f] for f in ("ms_total", "ms_coarse", "ms_group", "ms_scan", "ms_merge")])
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / a.steps
m = np.mean(np.array(acc), axis=0)
print(json.dumps({"workload": [a.n, a.d, a.nlist, a.nprobe, a.nq, a.k], "steps": a.steps, "warmup": a.warmup,
                  "ms_per_step_wall": round(wall * 1e3, 4), "queries_per_s": round(a.nq / wall, 1),
                  "pipeline_ms": {"total": round(float(m[0]), 4), "coarse": round(float(m[1]), 4), "grouping": round(float(m[2]), 4),
                                  "list_rank": round(float(m[3]), 4), "select": round(float(m[4]), 4)},
                  "rank_mode": int(st["rank_mode"]), "scanned_vectors": int(st["scanned_vectors"]),
                  "search_launches_in_trace": a.warmup + a.steps}), flush=True)
shutil.rmtree(work, ignore_errors=True)
