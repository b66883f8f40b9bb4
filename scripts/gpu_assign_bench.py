#!/usr/bin/env python3
"""k-means assign at scale on the GPU: MFMA-filtered exact assign vs hierarchical (reference) mode."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "vector-indexer_amd")]
from vector_indexer_py import _native as N  # noqa: E402

dev = torch.device("cuda", 0)
L = N.lib()
PEAK_TF = 157.3


def run(n, d, k, mode, reps=3, env=None):
    g = torch.Generator(device=dev); g.manual_seed(42)
    X = torch.randn(n, d, generator=g, device=dev)
    Cn = X[torch.randperm(n, generator=g, device=dev)[:k]].contiguous()
    lab = torch.empty(n, dtype=torch.int32, device=dev)
    st = N.AssignStats()
    torch.cuda.synchronize()
    for key, val in (env or {}).items():
        os.environ[key] = val
    best = None
    for _ in range(reps):
        N.check(L.vi_assign_device(0, X.data_ptr(), n, d, Cn.data_ptr(), k, 42, mode, lab.data_ptr(), C.byref(st)))
        if best is None or st.ms_total < best[0]:
            best = (st.ms_total, st.ms_filter, st.ambiguous_rows, st.used_mfma)
    for key in (env or {}):
        os.environ.pop(key, None)
    flops = 2.0 * n * k * d
    tf = flops / (best[1] * 1e-3) / 1e12 if best[1] > 0 else 0.0
    print(f"n={n} d={d} k={k} mode={mode} env={env} total={best[0]:.2f} ms filter={best[1]:.2f} ms "
          f"amb={best[2]} mfma={best[3]} filter_TFLOPs={tf:.1f} ({tf / PEAK_TF:.1%} of f32 MFMA peak) "
          f"end-to-end {flops / (best[0] * 1e-3) / 1e12:.1f} TF-equiv", flush=True)
    return lab


for (n, d, k) in [(1_000_000, 128, 4096), (1_000_000, 128, 16384), (1_000_000, 64, 16384), (1_000_000, 96, 16384)]:
    a = run(n, d, k, 1)
    if n * k <= 5e9:
        b = run(n, d, k, 1, reps=1, env={"VI_NO_MFMA": "1"})
        print("   labels equal to exact-order scan:", bool((a == b).all()), flush=True)
    run(n, d, k, 0, reps=2)
if os.environ.get("BIG"):
    run(10_000_000, 128, 16384, 1, reps=2)
    run(10_000_000, 128, 16384, 0, reps=2)
