"""Faiss-style benchmark harness over vector_indexer_py — the methodology of the reference's
bench/faiss_bench_official/bench_all_ivf.py (readers :88-171, synthetic recipe :55-80, eval_setting :283-363, JSON +
Markdown output :488-535) and its adapter (vector_indexer_adapter.py:75-140), re-implemented on top of the MI355X
engine so that the reference's way of measuring can drive it:

    python -m vector_indexer_py.harness --n 100000 --d 128 --nq 1000 --k 100 --nprobe 1,2,4,8,16,32,64
    python -m vector_indexer_py.harness --xb-path sift_base.fvecs --xq-path sift_query.fvecs --gt-path sift_groundtruth.ivecs

(defaults = scripts/run_faiss_bench.sh:51-58: N=100 000, D=128, NQ=1 000, K=100, nprobe sweep, 2 s per setting.)
Ground truth for synthetic data comes from an exact float64 brute force here (the reference asks faiss's IndexFlatL2,
which is not installed offline).
"""
import argparse
import json
import os
import time
from typing import Optional, Tuple

import numpy as np


# ---- readers (bench_all_ivf.py:88-171) ---------------------------------------------------------------------------
def _read_xvecs(path: str, max_rows: Optional[int], what: str) -> np.ndarray:
    """records of [int32 width][width x 4-byte values] -> the (rows, width) payload as int32"""
    size = os.path.getsize(path)
    if size < 4:
        raise ValueError(f"Empty or invalid {what} file: {path}")
    width = int(np.fromfile(path, dtype=np.int32, count=1)[0])
    if width <= 0:
        raise ValueError(f"Invalid {what} file (record width {width}): {path}")
    rec = width + 1
    words = size // 4
    if max_rows is not None:
        words = min(words, int(max_rows) * rec)
    elif size % 4 != 0 or words % rec != 0:
        raise ValueError(f"Invalid {what} file (size not multiple of width+1): {path} (width={width}, ints={words})")
    raw = np.fromfile(path, dtype=np.int32, count=words)
    if raw.size % rec != 0:
        raise ValueError(f"Invalid {what} file (size not multiple of width+1): {path} (width={width}, ints={raw.size})")
    raw = raw.reshape(-1, rec)
    if (raw[:, 0] != width).any():
        raise ValueError(f"Invalid {what} file (records of different width): {path}")
    return raw[:, 1:]


def read_fvecs(path: str, max_rows: Optional[int] = None) -> np.ndarray:
    """.fvecs: repeated [int32 dim][float32 x dim] -> float32 (n, dim)"""
    return np.ascontiguousarray(_read_xvecs(path, max_rows, "fvecs").view(np.float32))


def read_ivecs(path: str, max_rows: Optional[int] = None) -> np.ndarray:
    """.ivecs: repeated [int32 k][int32 x k] -> int64 (n, k)"""
    return np.ascontiguousarray(_read_xvecs(path, max_rows, "ivecs").astype(np.int64))


def load_vectors(path: str, max_rows: Optional[int] = None) -> np.ndarray:
    if path.endswith(".npy"):
        arr = np.load(path, mmap_mode="r")
        return np.ascontiguousarray(np.asarray(arr[:max_rows] if max_rows is not None else arr, dtype=np.float32))
    if path.endswith(".fvecs"):
        return read_fvecs(path, max_rows)
    raise ValueError(f"Unsupported vector file type (expected .npy or .fvecs): {path}")


def load_groundtruth(path: str, max_rows: Optional[int] = None) -> np.ndarray:
    if path.endswith(".npy"):
        arr = np.load(path, mmap_mode="r")
        return np.ascontiguousarray(np.asarray(arr[:max_rows] if max_rows is not None else arr, dtype=np.int64))
    if path.endswith(".ivecs"):
        return read_ivecs(path, max_rows)
    raise ValueError(f"Unsupported ground truth file type (expected .npy or .ivecs): {path}")


# ---- data ---------------------------------------------------------------------------------------------------------
def exact_ground_truth(xb: np.ndarray, xq: np.ndarray, k: int) -> np.ndarray:
    """exact k nearest by squared L2 (float64 brute force, blocked); ties by lower index"""
    k = min(k, xb.shape[0])
    out = np.empty((xq.shape[0], k), dtype=np.int64)
    xbt = np.ascontiguousarray(xb.T, dtype=np.float64)  # (once: the cast is 8 N D bytes)
    bn = (xbt ** 2).sum(0)
    for s in range(0, xq.shape[0], 256):
        q = xq[s:s + 256].astype(np.float64)
        dist = (q ** 2).sum(1)[:, None] - 2.0 * (q @ xbt) + bn[None, :]
        part = np.argpartition(dist, k - 1, axis=1)[:, :k] if k < xb.shape[0] else np.tile(np.arange(k), (q.shape[0], 1))
        pd = np.take_along_axis(dist, part, axis=1)
        order = np.lexsort((part, pd), axis=1)
        out[s:s + 256] = np.take_along_axis(part, order, axis=1)
    return out


def load_local_data(xb_path: str, xq_path: str, gt_path: Optional[str], n: int, nq: int, k: int):
    """load_local_npy_data (bench_all_ivf.py:175-260): the files sliced to (n, nq, k); dimensions must agree; a ground
    truth with fewer than k columns is an error; one that names rows beyond the sliced xb (or none at all) is recomputed
    exactly"""
    xb, xq = load_vectors(xb_path, n), load_vectors(xq_path, nq)
    if xb.shape[1] != xq.shape[1]:
        raise ValueError(f"Dimension mismatch: xb has d={xb.shape[1]}, xq has d={xq.shape[1]}")
    gt = None
    if gt_path:
        gt = load_groundtruth(gt_path, xq.shape[0])
        if gt.shape[1] < k:
            raise ValueError(f"Ground truth has only {gt.shape[1]} neighbors per query, but k={k} was requested")
        gt = np.ascontiguousarray(gt[:xq.shape[0], :k])
        if gt.size and int(gt.max()) >= xb.shape[0]:
            gt = None  # (written for a larger base set: recompute for the slice, as the reference does)
    if gt is None:
        gt = exact_ground_truth(xb, xq, k)
    return xb, xq, gt


def synthetic_dataset(n: int, d: int, nq: int, k: int, seed: int = 42) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """bench_all_ivf.py:55-80: xb, then xq, from one default_rng(seed).standard_normal stream"""
    rng = np.random.default_rng(seed)
    xb = rng.standard_normal((n, d)).astype(np.float32)
    xq = rng.standard_normal((nq, d)).astype(np.float32)
    return xb, xq, exact_ground_truth(xb, xq, k)


# ---- the adapter + eval loop --------------------------------------------------------------------------------------
class FaissStyleAdapter:
    """the Faiss-looking face of an index: .d, .nprobe (settable), .search(xq, k) -> (D, I)
    (vector_indexer_adapter.py:75-140; the reference hops through an asyncio thread, the search here is synchronous)"""

    def __init__(self, vector_index, k: int = 100):
        self._idx, self._k, self._nprobe = vector_index, k, 1

    @property
    def d(self) -> int:
        return self._idx.dimension

    @property
    def nprobe(self) -> int:
        return self._nprobe

    @nprobe.setter
    def nprobe(self, value: int):
        self._nprobe = int(value)

    def search(self, xq: np.ndarray, k: int):
        return self._idx.search_sync(np.ascontiguousarray(xq, dtype=np.float32), k, self._nprobe)

    def __repr__(self):
        return f"FaissStyleAdapter(d={self.d}, nprobe={self.nprobe})"


def eval_setting(index, xq, gt, k, min_time, clock=time.time, verbose=True):
    """bench_all_ivf.py:283-363 (recall form): search the whole query set again and again until min_time seconds have
    passed; ms per query and QPS from the mean; recall@r = share of queries whose true nearest neighbour gt[:, 0] is
    among the first r results, r in (1, 10, 100) with r <= k"""
    nq = xq.shape[0]
    nrun = 0
    t0 = clock()
    while True:
        D, I = index.search(xq, k)
        nrun += 1
        t1 = clock()
        if t1 - t0 > min_time:
            break
    ms_per_query = (t1 - t0) * 1000.0 / nq / nrun
    res = {"ms_per_query": ms_per_query, "qps": 1000.0 / ms_per_query, "nrun": nrun, "recalls": {}}
    nn = gt[:, 0:1]
    for rank in (1, 10, 100):
        if rank <= k:
            res["recalls"][rank] = float((I[:, :rank] == nn).any(axis=1).sum() / float(nq))
    if verbose:
        print("  ".join("R@%-3d=%.4f" % rv for rv in res["recalls"].items()),
              "  %9.3f ms/q  %9.1f QPS  (nrun=%d)" % (ms_per_query, res["qps"], nrun))
    return res


def run(xb, xq, gt, k, nprobes, min_time, work_dir=None, nlist=0, verbose=True):
    """build + the nprobe sweep of one backend (bench_all_ivf.py:427-480) -> result dict"""
    import vector_indexer_py as vip
    t0 = time.time()
    idx = vip.build(xb, work_dir, nlist=nlist)
    build_s = time.time() - t0
    adapter = FaissStyleAdapter(idx, k)
    out = {"backend": "vector_indexer (MI355X engine)", "n": int(xb.shape[0]), "d": int(xb.shape[1]),
           "nlist": int(idx.num_centroids), "k": int(k), "build_time_s": build_s, "search_results": {}}
    for p in nprobes:
        adapter.nprobe = p
        if verbose:
            print("nprobe=%-4d" % p, end=" ")
        out["search_results"][f"nprobe={p}"] = eval_setting(adapter, xq, gt, k, min_time, verbose=verbose)
    return out


def save_results(all_results, output_dir):
    """faiss_bench_results.json + .md (bench_all_ivf.py:488-535)"""
    os.makedirs(output_dir, exist_ok=True)
    with open(os.path.join(output_dir, "faiss_bench_results.json"), "w") as f:
        json.dump(all_results, f, indent=2)
    with open(os.path.join(output_dir, "faiss_bench_results.md"), "w") as f:
        f.write("# IVF Benchmark Results (Faiss eval_setting methodology)\n\n"
                "- every setting runs until the minimum test duration, timing is the mean over the runs\n"
                "- R@r = share of queries whose true nearest neighbour is among the first r results\n\n")
        for r in all_results:
            f.write(f"## {r['backend']}\n\n- n={r['n']}, d={r['d']}, nlist={r['nlist']}, k={r['k']}\n"
                    f"- Build time: {r['build_time_s']:.2f}s\n\n"
                    "| nprobe | R@1 | R@10 | R@100 | ms/query | QPS |\n|--------|-----|------|-------|----------|-----|\n")
            for key, res in r["search_results"].items():
                rec = {int(a): b for a, b in res["recalls"].items()}
                cell = lambda x: f"{rec[x]:.4f}" if x in rec else "-"  # noqa: E731
                f.write(f"| {key.replace('nprobe=', '')} | {cell(1)} | {cell(10)} | {cell(100)} | "
                        f"{res['ms_per_query']:.3f} | {res['qps']:.1f} |\n")
            f.write("\n")


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--n", type=int, default=100_000)      # scripts/run_faiss_bench.sh:51-58
    ap.add_argument("--d", type=int, default=128)
    ap.add_argument("--nq", type=int, default=1_000)
    ap.add_argument("--k", type=int, default=100)
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--nprobe", default="1,2,4,8,16,32,64")
    ap.add_argument("--min-test-duration", type=float, default=2.0)
    ap.add_argument("--nlist", type=int, default=0, help="0 = the reference's calculate_num_clusters(n)")
    ap.add_argument("--xb-path"), ap.add_argument("--xq-path"), ap.add_argument("--gt-path")
    ap.add_argument("--max-rows", type=int, default=None, help="cap on the rows read from --xb-path (in addition to --n)")
    ap.add_argument("--output-dir", default="bench_results")
    ap.add_argument("--work-dir", default=None)
    a = ap.parse_args(argv)
    if a.xb_path:
        if not a.xq_path:
            raise SystemExit("--xb-path needs --xq-path")
        xb, xq, gt = load_local_data(a.xb_path, a.xq_path, a.gt_path, min(a.n, a.max_rows) if a.max_rows else a.n, a.nq, a.k)
    else:
        xb, xq, gt = synthetic_dataset(a.n, a.d, a.nq, a.k, a.seed)
    res = run(xb, xq, gt, a.k, [int(p) for p in a.nprobe.split(",")], a.min_test_duration, a.work_dir, a.nlist)
    save_results([res], a.output_dir)
    return res


if __name__ == "__main__":
    main()
