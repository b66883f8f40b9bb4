#!/usr/bin/env python3
"""Experiment (diagnostic only): real-valued lists far from the origin — the bench's SIFT-shaped data plus uniform noise —
searched with the ranking images taken about the origin (VI_CENTER=0) and about the mean of the stored vectors, under
every VI_RANK_APPROX setting; prints the phase times and checks that all settings return the same ids and distances."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vector-indexer_amd")]
import bench  # noqa: E402
import vector_indexer_py as vip  # noqa: E402

dev = torch.device("cuda", 0)
n, d, nlist, nq, k = int(os.environ.get("N", 1_000_000)), int(os.environ.get("D", 128)), int(os.environ.get("NLIST", 4096)), \
    int(os.environ.get("NQ", 10000)), int(os.environ.get("K", 10))
P = int(os.environ.get("P", 32))
kind = os.environ.get("DATA", "sift-noise")
if kind == "sift-noise":
    xb, xq = bench.make_dataset(n, d, nq, 42, dev)
    xb = xb + torch.rand_like(xb) * 0.5
    xq = xq + torch.rand_like(xq) * 0.5
else:  # N(offset, 1)
    g = torch.Generator(device=dev).manual_seed(7)
    off = float(os.environ.get("OFFSET", 0.0))
    xb = torch.randn((n, d), generator=g, device=dev) + off
    xq = torch.randn((nq, d), generator=g, device=dev) + off
work = f"/tmp/vi_center_exp_{n}_{d}_{nlist}_{kind}_{os.environ.get('OFFSET', '0')}"
if not os.path.exists(work + "/index/index.bin"):
    vip.build(xb.cpu().numpy(), work, nlist=nlist, now_secs=1_700_000_000)
D = torch.empty((nq, k), dtype=torch.float32, device=dev)
I = torch.empty((nq, k), dtype=torch.int64, device=dev)
ref = None
for centre in ("0", None):
    if centre is None:
        os.environ.pop("VI_CENTER", None)
    else:
        os.environ["VI_CENTER"] = centre
    index = vip.load(work + "/index", work + "/shards", d)
    index.enable_timing(True)
    for approx in (None, "0", "1", "2"):
        if approx is None:
            os.environ.pop("VI_RANK_APPROX", None)
        else:
            os.environ["VI_RANK_APPROX"] = approx
        acc = []
        for r in range(8):
            index.search_device(xq.data_ptr(), nq, k, P, D.data_ptr(), I.data_ptr(), 0)
            st = index.last_stats()
            if r >= 3:
                acc.append([st[f] for f in ("ms_total", "ms_coarse", "ms_group", "ms_scan", "ms_merge")])
        torch.cuda.synchronize()
        if ref is None:
            ref = (D.clone(), I.clone())
        same = bool(torch.equal(ref[0].view(torch.int32), D.view(torch.int32)) and torch.equal(ref[1], I))
        m = np.mean(np.array(acc), axis=0)
        print(f"centre={'auto' if centre is None else centre} approx={'auto' if approx is None else approx} mode={st['rank_mode']} "
              f"total={m[0]:.3f} coarse={m[1]:.3f} group={m[2]:.3f} rank={m[3]:.3f} select={m[4]:.3f} same={same} "
              f"items={st['scan_items']} tiles={st['filter_tile_blocks']} gq={st['group_queries']} scanned={st['scanned_vectors']}", flush=True)
    del index
