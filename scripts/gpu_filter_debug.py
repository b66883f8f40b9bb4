#!/usr/bin/env python3
"""Diagnostic: distribution of exact re-evaluations per query on the bench workload (MFMA path)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "vector-indexer_amd")]
os.environ["VI_FILTER"] = "1"
import bench  # noqa: E402
import vector_indexer_py as vip  # noqa: E402

dev = torch.device("cuda", 0)
n, d, nlist, nq, k = 1_000_000, 128, 4096, 10000, 10
xb, xq = bench.make_dataset(n, d, nq, 42, dev)
work = "/tmp/vi_scan_bench"
if not os.path.exists(work + "/index/index.bin"):
    vip.build(xb.cpu().numpy(), work, nlist=nlist, now_secs=1_700_000_000)
index = vip.load(work + "/index", work + "/shards", d)
index.enable_timing(True)


def evals(q0, q1, n_probe=16):
    m = q1 - q0
    D = torch.empty((m, k), dtype=torch.float32, device=dev)
    I = torch.empty((m, k), dtype=torch.int64, device=dev)
    sub = xq[q0:q1].contiguous()
    index.search_device(sub.data_ptr(), m, k, n_probe, D.data_ptr(), I.data_ptr(), 0)
    torch.cuda.synchronize()
    st = index.last_stats()
    return st["filter_rechecked"], st["filter_accepted"], st["scanned_vectors"]


per = []
for c in range(0, nq, 100):
    e, a, s = evals(c, c + 100)
    per.append((e, a, s, c))
per.sort(reverse=True)
print("chunks (evals, consults, scanned, q0): top", per[:5], "median", per[len(per) // 2])
e, a, s, c = per[0]
single = sorted(((evals(q, q + 1) + (q,)) for q in range(c, c + 100)), reverse=True)
print("heaviest chunk per query: top", single[:8], "median", single[50])
