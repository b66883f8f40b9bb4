#!/usr/bin/env python3
"""Generates tests/golden/readers/* in THIS container (the reference's Python never ships to the GPU box).

Small .fvecs / .ivecs / .npy files are written with numpy, then read back with the reference's own readers
(/root/reference/bench/faiss_bench_official/bench_all_ivf.py: _read_fvecs :88-117, _read_ivecs :120-143,
_load_vectors :146-157, _load_groundtruth :160-171) and its eval_setting (:283-363) is run on a deterministic fake
index; what they return is stored as the expected outputs of vector_indexer_py.harness.

    python tests/golden/make_reader_fixtures.py        (needs /root/reference; faiss is NOT needed)
"""
import importlib.util
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "readers")
REF = "/root/reference/bench/faiss_bench_official/bench_all_ivf.py"


def ref_module():
    sys.path.insert(0, os.path.dirname(REF))
    spec = importlib.util.spec_from_file_location("ref_bench_all_ivf", REF)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def write_xvecs(path, rows, dtype):
    rows = np.asarray(rows, dtype=dtype)
    n, w = rows.shape
    rec = np.empty((n, w + 1), dtype=np.int32)
    rec[:, 0] = w
    rec[:, 1:] = rows.view(np.int32)
    rec.tofile(path)


class FakeIndex:
    """returns, for query i, the ids (i*7 + j*3) % 50 — a fixed answer the recall arithmetic can be checked on"""

    def __init__(self):
        self.calls = 0

    def search(self, xq, k):
        self.calls += 1
        nq = xq.shape[0]
        I = (np.arange(nq)[:, None] * 7 + np.arange(k)[None, :] * 3) % 50
        return np.zeros((nq, k), dtype=np.float32), I.astype(np.int64)


def main():
    os.makedirs(OUT, exist_ok=True)
    m = ref_module()
    rng = np.random.default_rng(7)
    fv = rng.standard_normal((37, 12)).astype(np.float32)
    fv[3, 5] = np.float32(-0.0)
    fv[4, 0] = np.float32(1e-42)          # a denormal: the readers must move bits, not values
    iv = rng.integers(0, 2 ** 31 - 1, size=(37, 9)).astype(np.int32)
    write_xvecs(os.path.join(OUT, "a.fvecs"), fv, np.float32)
    write_xvecs(os.path.join(OUT, "a.ivecs"), iv, np.int32)
    np.save(os.path.join(OUT, "a.npy"), fv.astype(np.float64))          # the .npy loader casts to f32
    np.save(os.path.join(OUT, "gt.npy"), iv.astype(np.int32))           # ... and ground truth to i64
    exp = {}
    for rows in (37, 10, 1):
        exp[f"fvecs_{rows}"] = m._read_fvecs(os.path.join(OUT, "a.fvecs"), max_rows=rows)
        exp[f"ivecs_{rows}"] = m._read_ivecs(os.path.join(OUT, "a.ivecs"), max_rows=rows)
        exp[f"npy_{rows}"] = m._load_vectors(os.path.join(OUT, "a.npy"), max_rows=rows)
        exp[f"gtnpy_{rows}"] = m._load_groundtruth(os.path.join(OUT, "gt.npy"), max_rows=rows)
    exp["load_fvecs"] = m._load_vectors(os.path.join(OUT, "a.fvecs"), max_rows=37)
    exp["load_ivecs"] = m._load_groundtruth(os.path.join(OUT, "a.ivecs"), max_rows=37)
    np.savez(os.path.join(OUT, "expected.npz"), **exp)
    # eval_setting on the fake index: recalls at k = 100, 10 and 5
    xq = np.zeros((40, 4), dtype=np.float32)
    gt = ((np.arange(40)[:, None] * 5 + np.arange(3)[None, :]) % 50).astype(np.int64)
    ev = {}
    for k in (100, 10, 5):
        r = m.eval_setting(FakeIndex(), xq, gt, k, False, 0.0)
        ev[str(k)] = {"recalls": {str(a): float(b) for a, b in r["recalls"].items()}, "keys": sorted(r.keys())}
    # the synthetic recipe (:67-69): checksums of the first draws of default_rng(42)
    rng = np.random.default_rng(42)
    xb = rng.standard_normal((1000, 16)).astype(np.float32)
    xq2 = rng.standard_normal((10, 16)).astype(np.float32)
    ev["synthetic"] = {"xb_sum_bits": int(np.float64(xb.astype(np.float64).sum()).view(np.uint64)),
                       "xq_first_bits": [int(v) for v in xq2[0].view(np.uint32)]}
    json.dump(ev, open(os.path.join(OUT, "eval_setting.json"), "w"), indent=1)
    print("wrote", sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()
