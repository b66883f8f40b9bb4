#!/usr/bin/env python3
"""Experiment harness (diagnostic only): phase times of the MFMA search pipeline on the bench workload under the
ablation knobs of the rank kernel (VI_FILTER_XMODE bits: 1 tiles not restaged, 2 no ranking epilogue, 8 no block-record
store, 16 no b1 insertion) and the other environment knobs.  Usage: gpu_rank_exp.py [nprobe ...]; EXP=name picks one
setting only (for rocprofv3 --pmc runs)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "vector-indexer_amd")]
import bench  # noqa: E402
import vector_indexer_py as vip  # noqa: E402

dev = torch.device("cuda", 0)
n, d, nlist, nq, k = int(os.environ.get("N", 1_000_000)), int(os.environ.get("D", 128)), int(os.environ.get("NLIST", 4096)), \
    int(os.environ.get("NQ", 10000)), int(os.environ.get("K", 10))
xb, xq = bench.make_dataset(n, d, nq, 42, dev)
if os.environ.get("REAL") == "1":  # real-valued data: bf16 x 3 ranking
    xb = xb + torch.rand_like(xb) * 0.5
    xq = xq + torch.rand_like(xq) * 0.5
work = f"/tmp/vi_rank_exp_{n}_{d}_{nlist}_{os.environ.get('REAL', '0')}"
if not os.path.exists(work + "/index/index.bin"):
    vip.build(xb.cpu().numpy(), work, nlist=nlist, now_secs=1_700_000_000)
index = vip.load(work + "/index", work + "/shards", d)
index.enable_timing(True)
D = torch.empty((nq, k), dtype=torch.float32, device=dev)
I = torch.empty((nq, k), dtype=torch.int64, device=dev)


def run(label, n_probe, reps=8, **env):
    for key, val in env.items():
        os.environ[key] = str(val)
    acc = {}
    for r in range(reps):
        index.search_device(xq.data_ptr(), nq, k, n_probe, D.data_ptr(), I.data_ptr(), 0)
        st = index.last_stats()
        if r >= 3:
            for f in ("ms_total", "ms_coarse", "ms_group", "ms_scan", "ms_merge"):
                acc.setdefault(f, []).append(st[f])
    for key in env:
        os.environ.pop(key, None)
    m = {f[3:]: round(float(np.mean(v)), 3) for f, v in acc.items()}
    print(f"{label:30s} P={n_probe:3d} {m} items={st['scan_items']} tiles={st['filter_tile_blocks']} "
          f"fill={st['scanned_vectors'] / max(1, st['filter_tile_blocks'] * 64 * max(1, st['group_queries'])):.2f} "
          f"mode={st['rank_mode']} gq={st['group_queries']}", flush=True)


SETTINGS = [
    ("default", {}),
    ("tiles not restaged (x1)", {"VI_FILTER_XMODE": 1}),
    ("no epilogue (x2)", {"VI_FILTER_XMODE": 2}),
    ("no brec store (x8)", {"VI_FILTER_XMODE": 8}),
    ("no restage, no epilogue (x3)", {"VI_FILTER_XMODE": 3}),
    ("no restage, no store (x9)", {"VI_FILTER_XMODE": 9}),
    ("gq 32", {"VI_FILTER_GQ": 32}),
    ("bf16x3 lo planes", {"VI_FILTER_HI_ONLY": 0}),
    ("select: no exact (s1)", {"VI_SELECT_XMODE": 1}),
    ("select: no stage 2 (s2)", {"VI_SELECT_XMODE": 2}),
    ("select: no stage 1b (s4)", {"VI_SELECT_XMODE": 4}),
    ("select: stage 1a only (s7)", {"VI_SELECT_XMODE": 7}),
    ("segb 16", {"VI_FILTER_SEGB": 16}),
    ("segb 64", {"VI_FILTER_SEGB": 64}),
    ("x16 prologue only", {"VI_FILTER_XMODE": 16}),
    ("x48 prologue without gather", {"VI_FILTER_XMODE": 48}),
    ("gq256", {"VI_STREAM_GQ": 256}),
    ("gq256 x1", {"VI_STREAM_GQ": 256, "VI_FILTER_XMODE": 1}),
    ("gq256 x16", {"VI_STREAM_GQ": 256, "VI_FILTER_XMODE": 16}),
    ("gq256 segb 16", {"VI_STREAM_GQ": 256, "VI_FILTER_SEGB": 16}),
    ("gq256 segb 64", {"VI_STREAM_GQ": 256, "VI_FILTER_SEGB": 64}),
    ("exact from f32", {"VI_EXACT_BF16": 0, "VI_EXACT_U8": 0}),
    ("coarse records by block", {"VI_COARSE_QMAJOR": 0}),
    ("coarse rows from blocks", {"VI_COARSE_ROWS": 0}),
    ("s x1 tiles not re-read", {"VI_FILTER_XMODE": 1}),
    ("s x2 no epilogue", {"VI_FILTER_XMODE": 2}),
    ("s x3", {"VI_FILTER_XMODE": 3}),
    ("s x8 no record store", {"VI_FILTER_XMODE": 8}),
    ("s x16 no multiply", {"VI_FILTER_XMODE": 16}),
    ("s x32 no gather", {"VI_FILTER_XMODE": 32}),
    ("s x43", {"VI_FILTER_XMODE": 43}),
    ("gq256 segb 128", {"VI_STREAM_GQ": 256, "VI_FILTER_SEGB": 128}),
    ("gq128 segb 64", {"VI_STREAM_GQ": 128, "VI_FILTER_SEGB": 64}),
    ("gq128 segb 128", {"VI_STREAM_GQ": 128, "VI_FILTER_SEGB": 128}),
    ("gq256 1 wg/cu", {"VI_STREAM_GQ": 256, "VI_STREAM_WGS_PER_CU": 1}),
    ("gq256 4 wg/cu", {"VI_STREAM_GQ": 256, "VI_STREAM_WGS_PER_CU": 4}),
    ("gq128 1 wg/cu", {"VI_STREAM_GQ": 128, "VI_STREAM_WGS_PER_CU": 1}),
    ("gq256 x2 no epilogue", {"VI_STREAM_GQ": 256, "VI_FILTER_XMODE": 2}),
    ("gq256 x3", {"VI_STREAM_GQ": 256, "VI_FILTER_XMODE": 3}),
    ("gq256 x64 static", {"VI_STREAM_GQ": 256, "VI_FILTER_XMODE": 64}),
    ("gq256 x32 no gather", {"VI_STREAM_GQ": 256, "VI_FILTER_XMODE": 32}),
    ("gq256 x96", {"VI_STREAM_GQ": 256, "VI_FILTER_XMODE": 96}),
    ("gq256 x80", {"VI_STREAM_GQ": 256, "VI_FILTER_XMODE": 80}),
    ("gq128", {"VI_STREAM_GQ": 128}),
    ("old kernel", {"VI_RANK_STREAM": 0}),
    ("run 1", {"VI_ITEM_RUN": 1}),
    ("run 4", {"VI_ITEM_RUN": 4}),
    ("run 16", {"VI_ITEM_RUN": 16}),
    ("run 32", {"VI_ITEM_RUN": 32}),
    ("run 16 no epilogue", {"VI_ITEM_RUN": 16, "VI_FILTER_XMODE": 2}),
    ("run 16 segb 64", {"VI_ITEM_RUN": 16, "VI_FILTER_SEGB": 64}),
]
if os.environ.get("SET"):  # SET=a,b,c: only these settings (prefix match)
    want = os.environ["SET"].split(",")
    SETTINGS = [s for s in SETTINGS if any(s[0].startswith(w) for w in want)]
if os.environ.get("AB"):  # AB=a,b: alternate two settings three times (same process, same box)
    names = os.environ["AB"].split(",")
    ab = [s for s in SETTINGS if s[0] in names]
    SETTINGS = ab * 3
only = os.environ.get("EXP")
probes = [int(a) for a in sys.argv[1:]] or [16, 32]
for p in probes:
    for name, env in SETTINGS:
        if only and only != name.split(" (")[0] and only != name:
            continue
        run(name, p, **env)
