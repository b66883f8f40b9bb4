#!/usr/bin/env python3
"""BASELINE config C3: mini-batch k-means train N=1e7 D=128 k=16384 through the C ABI (host matrix in, centroids
and labels out), reference-compat (hierarchical final assign) and exact final assign."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "vector-indexer_amd")]
import vector_indexer_py as vip  # noqa: E402

n, d, k = int(os.environ.get("N", 10_000_000)), 128, 16384
rng = np.random.default_rng(42)
t0 = time.time()
X = np.empty((n, d), dtype=np.float32)
for s in range(0, n, 1_000_000):
    X[s:s + 1_000_000] = rng.standard_normal((min(1_000_000, n - s), d), dtype=np.float32)
print(f"generated {n}x{d} in {time.time() - t0:.1f}s", flush=True)
for mode, name in ((vip.VI_ASSIGN_REFERENCE, "reference (hierarchical final assign)"), (vip.VI_ASSIGN_EXACT, "exact final assign")):
    t0 = time.time()
    C, labels, iters = vip.kmeans_mini_batch(X, k, 20, None, 42, mode)
    dt = time.time() - t0
    print(f"mini-batch k-means {name}: {dt:.2f}s, {iters} iterations, labels used {len(np.unique(labels))}", flush=True)
