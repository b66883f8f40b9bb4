#!/usr/bin/env python3
"""Diagnostic: how the scan work of the bench workload is spread over the inverted lists."""
import os
import sys

import numpy as np
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
C, c2s = index.centroids()
C = torch.from_numpy(C).to(dev)
cn = (C * C).sum(1)


def nearest(x, p):
    out = []
    for s in range(0, x.shape[0], 65536):
        blk = x[s:s + 65536]
        dist = cn[None, :] - 2.0 * (blk @ C.T)
        out.append(torch.topk(dist, p, dim=1, largest=False).indices)
    return torch.cat(out)


lens = torch.bincount(nearest(xb, 1).flatten(), minlength=C.shape[0]).double()
probes = torch.bincount(nearest(xq, 16).flatten(), minlength=C.shape[0]).double()
work_l = lens * probes
tot = work_l.sum()
order = torch.argsort(work_l, descending=True)
print("lists", C.shape[0], "scanned/q", float(tot / nq))
for i in order[:8].tolist():
    print(f"list {i}: len {int(lens[i])} probes {int(probes[i])} share {float(work_l[i] / tot):.4f}")
for N in (2, 4, 8):
    for name, w in (("len^2", lens * lens), ("len", lens), ("true work", work_l)):
        load = np.zeros(N)
        share = np.zeros(N)
        for i in torch.argsort(w, descending=True).tolist():
            r = int(np.argmin(load))
            load[r] += float(w[i])
            share[r] += float(work_l[i])
        print(f"N={N} LPT by {name}: max share {share.max() / share.sum():.3f} (ideal {1 / N:.3f})")
