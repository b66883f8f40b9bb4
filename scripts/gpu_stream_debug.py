#!/usr/bin/env python3
"""Diagnostic: streaming rank kernel against the block-synchronous one and the oracle on small random indexes."""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "vector-indexer_amd")]
import oracle_lib as O  # noqa: E402
import vector_indexer_py as vip  # noqa: E402


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


rng = np.random.default_rng(7)
cases = [(200, 8, 1, 1, 50, False), (3000, 8, 5, 3, 8, False), (3000, 32, 40, 10, 8, True), (20000, 128, 300, 10, 16, True),
         (20000, 64, 300, 10, 16, False)]
for n, d, nq, k, P, integer in cases:
    X = rng.integers(0, 200, size=(n, d)).astype(np.float32) if integer else rng.standard_normal((n, d)).astype(np.float32)
    Q = X[rng.integers(0, n, nq)] + (0 if integer else 0.01 * rng.standard_normal((nq, d))).astype(np.float32) if not integer else X[rng.integers(0, n, nq)]
    Q = np.ascontiguousarray(Q, dtype=np.float32)
    with tempfile.TemporaryDirectory() as tmp:
        orc = O.OracleIndex.build(X, tmp + "/index", tmp + "/shards", nlist=0, seed=42)
        gpu = vip.load(tmp + "/index", tmp + "/shards", d)
        rc, Do, Io = orc.search_batch(Q, k, P)
        for stream in ("0", "1", "256"):
            os.environ["VI_RANK_STREAM"] = "0" if stream == "0" else "1"
            os.environ["VI_STREAM_GQ"] = "256" if stream == "256" else "128"
            Dg, Ig = gpu.search_sync(Q, k, P)
            bad = np.nonzero((Ig != Io).any(axis=1) | (bits(Dg) != bits(Do)).any(axis=1))[0]
            st = gpu.last_stats()
            print(f"n={n} d={d} nq={nq} k={k} P={P} int={integer} lists={gpu.num_centroids} stream={stream}: {bad.size} bad, mode={st['rank_mode']} items={st['scan_items']}", flush=True)
            if bad.size:
                b = bad[0]
                print("   first", b, "gpu", Ig[b], Dg[b], "oracle", Io[b], Do[b], flush=True)
