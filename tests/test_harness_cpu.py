"""vector_indexer_py.harness (Faiss-style bench: readers, eval_setting, adapter, JSON+MD) against fixtures produced by
the reference's own Python harness (tests/golden/make_reader_fixtures.py, run in the build container)."""
import json
import os

import numpy as np
import pytest

from vector_indexer_py import harness as H

R = os.path.join(os.path.dirname(__file__), "golden", "readers")
EXP = np.load(os.path.join(R, "expected.npz"))


def same_bits(a, b):
    return a.dtype == b.dtype and a.shape == b.shape and (a.view(np.uint8) == b.view(np.uint8)).all()


@pytest.mark.parametrize("rows", [37, 10, 1])
def test_readers_match_the_reference_readers(rows):
    assert same_bits(H.read_fvecs(os.path.join(R, "a.fvecs"), rows), EXP[f"fvecs_{rows}"])
    assert same_bits(H.read_ivecs(os.path.join(R, "a.ivecs"), rows), EXP[f"ivecs_{rows}"])
    assert same_bits(H.load_vectors(os.path.join(R, "a.npy"), rows), EXP[f"npy_{rows}"])
    assert same_bits(H.load_groundtruth(os.path.join(R, "gt.npy"), rows), EXP[f"gtnpy_{rows}"])


def test_loaders_dispatch_on_the_extension_and_read_whole_files():
    assert same_bits(H.load_vectors(os.path.join(R, "a.fvecs")), EXP["load_fvecs"])       # max_rows=None works here
    assert same_bits(H.load_groundtruth(os.path.join(R, "a.ivecs")), EXP["load_ivecs"])
    assert H.read_fvecs(os.path.join(R, "a.fvecs")).flags["C_CONTIGUOUS"]
    with pytest.raises(ValueError):
        H.load_vectors("x.bin")
    with pytest.raises(ValueError):
        H.load_groundtruth("x.fvecs")


def test_truncated_and_ragged_files_are_rejected(tmp_path):
    raw = np.fromfile(os.path.join(R, "a.fvecs"), dtype=np.int32)
    p = str(tmp_path / "cut.fvecs")
    raw[:-3].tofile(p)
    with pytest.raises(ValueError):
        H.read_fvecs(p)
    bad = raw.copy()
    bad[13] = 11            # second record claims another width
    p2 = str(tmp_path / "ragged.fvecs")
    bad.tofile(p2)
    with pytest.raises(ValueError):
        H.read_fvecs(p2)
    open(str(tmp_path / "empty.ivecs"), "wb").close()
    with pytest.raises(ValueError):
        H.read_ivecs(str(tmp_path / "empty.ivecs"))


class FakeIndex:   # the index of make_reader_fixtures.py
    def __init__(self):
        self.calls = 0

    def search(self, xq, k):
        self.calls += 1
        nq = xq.shape[0]
        I = (np.arange(nq)[:, None] * 7 + np.arange(k)[None, :] * 3) % 50
        return np.zeros((nq, k), dtype=np.float32), I.astype(np.int64)


def test_eval_setting_recalls_and_timing_loop():
    gold = json.load(open(os.path.join(R, "eval_setting.json")))
    xq = np.zeros((40, 4), dtype=np.float32)
    gt = ((np.arange(40)[:, None] * 5 + np.arange(3)[None, :]) % 50).astype(np.int64)
    for k in (100, 10, 5):
        r = H.eval_setting(FakeIndex(), xq, gt, k, 0.0, verbose=False)
        assert {str(a): b for a, b in r["recalls"].items()} == gold[str(k)]["recalls"]
        assert sorted(r.keys()) == gold[str(k)]["keys"]
    # the loop runs until min_time has passed on the given clock and averages over the runs
    ticks = iter([0.0, 0.4, 0.8, 1.2, 1.6, 2.5])
    idx = FakeIndex()
    r = H.eval_setting(idx, xq, gt, 10, 2.0, clock=lambda: next(ticks), verbose=False)
    assert idx.calls == r["nrun"] == 5
    assert abs(r["ms_per_query"] - 2500.0 / 40 / 5) < 1e-9 and abs(r["qps"] - 1000.0 / r["ms_per_query"]) < 1e-9


def test_synthetic_recipe_and_exact_ground_truth():
    gold = json.load(open(os.path.join(R, "eval_setting.json")))["synthetic"]
    xb, xq, gt = H.synthetic_dataset(1000, 16, 10, 5, seed=42)
    assert int(np.float64(xb.astype(np.float64).sum()).view(np.uint64)) == gold["xb_sum_bits"]
    assert [int(v) for v in xq[0].view(np.uint32)] == gold["xq_first_bits"]
    d = ((xq[:, None, :].astype(np.float64) - xb[None, :, :].astype(np.float64)) ** 2).sum(2)
    assert (gt == np.argsort(d, axis=1, kind="stable")[:, :5]).all()


def test_adapter_and_result_files(tmp_path):
    class Idx:
        dimension = 4

        def search_sync(self, xq, k, n_probe):
            self.seen = (xq.dtype, xq.flags["C_CONTIGUOUS"], k, n_probe)
            return np.zeros((xq.shape[0], k), np.float32), np.zeros((xq.shape[0], k), np.int64)
    idx = Idx()
    a = H.FaissStyleAdapter(idx, k=100)
    assert a.d == 4 and a.nprobe == 1
    a.nprobe = 16
    a.search(np.zeros((6, 4), dtype=np.float64)[::2], 7)
    assert idx.seen == (np.float32, True, 7, 16)
    res = {"backend": "b", "n": 10, "d": 4, "nlist": 3, "k": 10, "build_time_s": 1.5,
           "search_results": {"nprobe=1": {"ms_per_query": 0.5, "qps": 2000.0, "nrun": 3, "recalls": {1: 0.25, 10: 0.5}}}}
    H.save_results([res], str(tmp_path))
    back = json.load(open(tmp_path / "faiss_bench_results.json"))
    assert back[0]["search_results"]["nprobe=1"]["qps"] == 2000.0
    md = open(tmp_path / "faiss_bench_results.md").read()
    assert "| 1 | 0.2500 | 0.5000 | - | 0.500 | 2000.0 |" in md and "Build time: 1.50s" in md


def test_load_local_data_slices_and_validates(tmp_path):
    """load_local_npy_data (bench_all_ivf.py:175-260): n / nq / k slicing, dimension check, gt width check, gt recomputed
    when it names rows beyond the slice"""
    from vector_indexer_py import harness as H
    rng = np.random.default_rng(3)
    xb, xq = rng.standard_normal((300, 8)).astype(np.float32), rng.standard_normal((40, 8)).astype(np.float32)
    gt = H.exact_ground_truth(xb, xq, 10)
    np.save(tmp_path / "xb.npy", xb), np.save(tmp_path / "xq.npy", xq), np.save(tmp_path / "gt.npy", gt)
    np.save(tmp_path / "xq5.npy", xq[:, :5].copy()), np.save(tmp_path / "gt3.npy", gt[:, :3].copy())
    b, q, g = H.load_local_data(str(tmp_path / "xb.npy"), str(tmp_path / "xq.npy"), str(tmp_path / "gt.npy"), 300, 25, 7)
    assert b.shape == (300, 8) and q.shape == (25, 8) and (g == gt[:25, :7]).all()
    # sliced base set: the stored gt names rows >= 120 -> recomputed for the slice
    b, q, g = H.load_local_data(str(tmp_path / "xb.npy"), str(tmp_path / "xq.npy"), str(tmp_path / "gt.npy"), 120, 40, 5)
    assert b.shape == (120, 8) and g.max() < 120 and (g == H.exact_ground_truth(xb[:120], xq, 5)).all()
    with pytest.raises(ValueError, match="Dimension mismatch"):
        H.load_local_data(str(tmp_path / "xb.npy"), str(tmp_path / "xq5.npy"), None, 300, 40, 5)
    with pytest.raises(ValueError, match="only 3 neighbors"):
        H.load_local_data(str(tmp_path / "xb.npy"), str(tmp_path / "xq.npy"), str(tmp_path / "gt3.npy"), 300, 40, 5)
