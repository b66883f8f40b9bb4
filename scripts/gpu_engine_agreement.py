#!/usr/bin/env python3
"""Stress check on the bench workload: every engine / rank arithmetic returns the same (D, I) bits for all 10 000
queries, repeatedly (a race in the double-buffered tile pipeline would show up here)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "vector-indexer_amd")]
import bench  # noqa: E402
import vector_indexer_py as vip  # noqa: E402

dev = torch.device("cuda", 0)
n, d, nlist, nq, k = 1_000_000, 128, 4096, 10000, 10
xb, xq = bench.make_dataset(n, d, nq, 42, dev)
work = "/tmp/vi_scan_bench"
if not os.path.exists(work + "/index/index.bin"):
    vip.build(xb.cpu().numpy(), work, nlist=nlist, now_secs=1_700_000_000)
index = vip.load(work + "/index", work + "/shards", d)


def run(env, p):
    for key, val in env.items():
        os.environ[key] = val
    D = torch.empty((nq, k), dtype=torch.float32, device=dev)
    I = torch.empty((nq, k), dtype=torch.int64, device=dev)
    index.search_device(xq.data_ptr(), nq, k, p, D.data_ptr(), I.data_ptr(), 0)
    torch.cuda.synchronize()
    for key in env:
        os.environ.pop(key, None)
    return D.view(torch.int32).clone(), I.clone()


bad = 0
for p in (1, 16, 64):
    ref = run({"VI_FILTER": "0"}, p)
    for name, env in (("default", {}), ("bf16x3", {"VI_FILTER_HI_ONLY": "0"}), ("f32 mfma", {"VI_FILTER_BF16": "0"})):
        for rep in range(6 if name == "default" else 2):
            got = run(env, p)
            same = bool((got[0] == ref[0]).all()) and bool((got[1] == ref[1]).all())
            bad += 0 if same else 1
            if not same:
                print("MISMATCH", name, "nprobe", p, "rep", rep, int((got[1] != ref[1]).any(dim=1).sum()), "queries", flush=True)
print("engine agreement:", "OK" if bad == 0 else f"{bad} mismatching runs")
