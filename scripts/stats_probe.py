#!/usr/bin/env python3
"""Diagnostic: the select kernels' counters and stage clocks at the headline point (C2, nprobe 32, 10 000 queries).
    python scripts/stats_probe.py 2    coarse select: rows evaluated exactly, sampled stage clocks
    python scripts/stats_probe.py 3    list select: counters of every query (the atomics distort the clocks)
    python scripts/stats_probe.py 4    list select: counters and stage clocks of every 64th query"""
import os, sys
os.environ["VI_FILTER_STATS"] = sys.argv[1]
ROOT = "/root/repo"
sys.path[:0] = [ROOT, os.path.join(ROOT, "vector-indexer_amd")]
import torch, bench, tempfile, shutil
import vector_indexer_py as vip
dev = torch.device("cuda", 0)
xb, xq = bench.make_dataset(1_000_000, 128, 10000, 42, dev)
work = "/tmp/vi_stats_probe"
shutil.rmtree(work, ignore_errors=True)
idx = vip.build(xb.cpu().numpy(), work, nlist=4096, now_secs=1_700_000_000)
idx.enable_timing(True)
D = torch.empty((10000, 10), dtype=torch.float32, device=dev); I = torch.empty((10000, 10), dtype=torch.int64, device=dev)
for r in range(3):
    idx.search_device(xq.data_ptr(), 10000, 10, 32, D.data_ptr(), I.data_ptr(), 0)
st = idx.last_stats()
print("STATS", sys.argv[1], {k: st[k] for k in ("filter_rechecked", "filter_accepted", "ms_coarse", "ms_merge")})
