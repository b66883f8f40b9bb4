#!/usr/bin/env python3
"""Experiment harness: time the search phases under tuning knobs (VI_SEG_BLOCKS, VI_FORCE_QG,
VI_NO_SELECT) on the bench workload.  Diagnostic only."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "vector-indexer_amd")]
import bench  # noqa: E402
import vector_indexer_py as vip  # noqa: E402

dev = torch.device("cuda", 0)
n, d, nlist, nq, k = 1_000_000, 128, 4096, int(os.environ.get("NQ", 10000)), 10
xb, xq = bench.make_dataset(n, d, nq, 42, dev)
work = "/tmp/vi_scan_bench"
if not os.path.exists(work + "/index/index.bin"):
    vip.build(xb.cpu().numpy(), work, nlist=nlist, now_secs=1_700_000_000)
index = vip.load(work + "/index", work + "/shards", d)
index.enable_timing(True)
D = torch.empty((nq, k), dtype=torch.float32, device=dev)
I = torch.empty((nq, k), dtype=torch.int64, device=dev)


def run(label, n_probe, reps=6, **env):
    for key, val in env.items():
        os.environ[key] = str(val)
    acc = {}
    for r in range(reps):
        index.search_device(xq.data_ptr(), nq, k, n_probe, D.data_ptr(), I.data_ptr(), 0)
        st = index.last_stats()
        if r >= 2:
            for f in ("ms_total", "ms_coarse", "ms_group", "ms_scan", "ms_merge"):
                acc.setdefault(f, []).append(st[f])
    for key in env:
        os.environ.pop(key, None)
    m = {f: round(float(np.mean(v)), 3) for f, v in acc.items()}
    lane_ops = st["scanned_vectors"] * d * 3
    print(f"{label:34s} nprobe={n_probe:3d} {m} items={st['scan_items']} pairs={nq * n_probe} "
          f"scanned/q={st['scanned_vectors'] / nq:.0f} valu_frac={lane_ops / (m['ms_scan'] * 1e-3) / 78.6e12:.3f} "
          f"tile_blocks={st['filter_tile_blocks']} fill={st['scanned_vectors'] / max(1, st['filter_tile_blocks'] * 2048):.2f} "
          f"rechecked={st['filter_rechecked']} accepted={st['filter_accepted']} fallback={st['fallback_queries']}", flush=True)


for p in (16,):
    run("default", p)
    run("no epilogue", p, VI_FILTER_XMODE=2)
    run("no brec store", p, VI_FILTER_XMODE=8)
    run("segb 16", p, VI_FILTER_SEGB=16)
    run("segb 64", p, VI_FILTER_SEGB=64)
