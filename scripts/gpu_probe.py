#!/usr/bin/env python3
"""Phase timings of the search path on the GPU box (diagnostic, not a benchmark)."""
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "vector-indexer_amd")]
import oracle_lib as O  # noqa: E402
import vector_indexer_py as vip  # noqa: E402


def t(label, fn):
    t0 = time.time()
    r = fn()
    print(f"{label:40s} {time.time() - t0:8.3f} s", flush=True)
    return r


print("cpus", os.cpu_count(), "usable", O.usable_cpus(), "omp", os.environ.get("OMP_NUM_THREADS"), flush=True)
rng = np.random.default_rng(0)
n, d = int(sys.argv[1]) if len(sys.argv) > 1 else 20000, int(sys.argv[2]) if len(sys.argv) > 2 else 64
X = rng.standard_normal((n, d)).astype(np.float32)
Q = rng.standard_normal((2000, d)).astype(np.float32)
tmp = tempfile.mkdtemp()
orc = t("oracle build", lambda: O.OracleIndex.build(X, tmp + "/index", tmp + "/shards"))
gpu = t("gpu load", lambda: vip.load(tmp + "/index", tmp + "/shards", d))
gpu.enable_timing(True)
for nq in (1, 64, 2000):
    for rep in range(2):
        r = t(f"gpu search nq={nq} rep={rep}", lambda: gpu.search_sync(Q[:nq], 10, 8))
    print("   stats", {k: (round(v, 3) if isinstance(v, float) else v) for k, v in gpu.last_stats().items()}, flush=True)
rc, Do, Io = t("oracle search 2000", lambda: orc.search_batch(Q, 10, 8))
Dg, Ig = gpu.search_sync(Q, 10, 8)
print("ids equal", bool((Ig == Io).all()), "dist bits equal", bool((Dg.view(np.uint32) == Do.view(np.uint32)).all()))
tmp2 = tempfile.mkdtemp()
g2 = t("gpu build (kmeans on GPU)", lambda: vip.build(X, tmp2, now_secs=1_700_000_000))
orc2 = O.OracleIndex.load(tmp2 + "/index", tmp2 + "/shards")
import filecmp
same = all(filecmp.cmp(os.path.join(tmp, "shards", f), os.path.join(tmp2, "shards", f), shallow=False)
           for f in os.listdir(os.path.join(tmp, "shards")))
print("gpu-built shard files identical to oracle-built:", same,
      "index.bin:", filecmp.cmp(tmp + "/index/index.bin", tmp2 + "/index/index.bin", shallow=False))
