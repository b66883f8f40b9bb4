"""vector_indexer_py — ctypes mirror of the reference's Python module, on top of libvi_amd.so.

Reference contract (bindings/python/src/lib.rs:120-325 and
bindings/python/python/vector_indexer_py/__init__.py):

    build(xb, work_dir=None) -> VectorIndex       external_id = row index, dirs <work_dir>/{index,shards}
    load(index_dir, shards_dir, dimension) -> VectorIndex
    suggest_nlist(n) -> int
    VectorIndex.search(xq, k, n_probe)   (coroutine) -> (D f32[nq,k], I i64[nq,k])   +inf / -1 padded
    VectorIndex.search_sync(xq, k, n_probe)
    VectorIndex.dimension

Errors surface as RuntimeError, as PyO3's PyRuntimeError does.  There is NO CPU fallback: if
libvi_amd.so is missing or no MI355X is visible, build/load/search raise.
"""
import asyncio
import ctypes as C
import os
import tempfile
from typing import Optional, Tuple

import numpy as np

from . import _native
from ._native import (ViError, lib, VI_ASSIGN_EXACT, VI_ASSIGN_REFERENCE, VI_ORDER_LANES, VI_ORDER_SCALAR)

__all__ = ["build", "load", "suggest_nlist", "VectorIndex", "ViError", "kmeans_mini_batch", "kmeans_parallel",
           "assign", "l2sq_pairs", "VI_ASSIGN_EXACT", "VI_ASSIGN_REFERENCE", "VI_ORDER_LANES", "VI_ORDER_SCALAR"]


def suggest_nlist(n: int) -> int:
    """lib.rs:308-315 (mirrors src/utils.rs:9-16)."""
    return int(lib().vi_calculate_num_clusters(int(n)))


class VectorIndex:
    """Index handle (PyVectorIndex + the async wrapper of the reference)."""

    def __init__(self, handle, dimension, keep=None):
        self._h = handle
        self._dimension = int(dimension)
        self._keep = keep

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and lib is not None:  # (module globals are already torn down at interpreter exit)
            try:
                lib().vi_indexer_free(h)
            except Exception:
                pass

    @property
    def dimension(self) -> int:
        return self._dimension

    @property
    def num_centroids(self) -> int:
        return int(lib().vi_indexer_num_centroids(self._h))

    @property
    def num_vectors(self) -> int:
        return int(lib().vi_indexer_num_vectors(self._h))

    @property
    def num_shards(self) -> int:
        return int(lib().vi_indexer_num_shards(self._h))

    def centroids(self):
        k, d = self.num_centroids, self._dimension
        cent = np.zeros((k, d), dtype=np.float32)
        c2s = np.zeros(k, dtype=np.uint64)
        _native.check(lib().vi_indexer_centroids(self._h, _native.ptr(cent), _native.ptr(c2s)))
        return cent, c2s

    # PyVectorIndex.search_blocking (lib.rs:123-203)
    def search_sync(self, xq, k: int, n_probe: int, include_vectors: bool = False):
        xq = np.asarray(xq)
        if xq.ndim != 2:
            raise RuntimeError("Query array must be 2-dimensional")
        if xq.shape[1] != self._dimension:  # lib.rs:133-138
            raise RuntimeError(f"Query dimension {xq.shape[1]} doesn't match index dimension {self._dimension}")
        xq = np.ascontiguousarray(xq, dtype=np.float32)
        nq = xq.shape[0]
        k = int(k)
        D = np.full((nq, max(k, 0)), np.inf, dtype=np.float32)
        I = np.full((nq, max(k, 0)), -1, dtype=np.int64)
        V = np.zeros((nq, k, self._dimension), dtype=np.float32) if include_vectors else None
        kout = C.c_uint64(0)
        _native.check(lib().vi_indexer_search(self._h, _native.ptr(xq), nq, xq.shape[1], k, int(n_probe),
                                              _native.ptr(D), _native.ptr(I), _native.ptr(V), None,
                                              C.byref(kout)))
        if kout.value != k:  # k was clamped to max_k (api.rs:189): results occupy the first kout columns
            kk = kout.value
            D2 = np.full((nq, k), np.inf, dtype=np.float32)
            I2 = np.full((nq, k), -1, dtype=np.int64)
            D2[:, :kk] = D.reshape(-1)[: nq * kk].reshape(nq, kk)
            I2[:, :kk] = I.reshape(-1)[: nq * kk].reshape(nq, kk)
            D, I = D2, I2
        return (D, I, V) if include_vectors else (D, I)

    async def search(self, xq, k: int, n_probe: int) -> Tuple[np.ndarray, np.ndarray]:
        loop = asyncio.get_event_loop()
        return await loop.run_in_executor(None, self.search_sync, xq, k, n_probe)

    def enable_timing(self, on=True):
        """True / 1: HIP events at every phase boundary; 2: around the list-rank kernel only (ms_scan); False / 0: none"""
        lib().vi_indexer_enable_timing(self._h, 2 if on == 2 and on is not True else (1 if on else 0))

    def last_stats(self) -> dict:
        st = _native.SearchStats()
        _native.check(lib().vi_indexer_last_stats(self._h, C.byref(st)))
        return {f: getattr(st, f) for f, _ in st._fields_}

    def last_stat(self, field: str):
        """one field of last_stats() without building the dict (a timed loop reads ms_scan after every step)"""
        st = self.__dict__.get("_stat_buf")
        if st is None:
            st = self.__dict__["_stat_buf"] = _native.SearchStats()
        _native.check(lib().vi_indexer_last_stats(self._h, C.byref(st)))
        return getattr(st, field)

    def build_stats(self) -> dict:
        st = _native.BuildStats()
        _native.check(lib().vi_indexer_last_build_stats(self._h, C.byref(st)))
        return {f: getattr(st, f) for f, _ in st._fields_}

    # device-pointer search used by the multi-GPU path (torch tensors on this GPU)
    def search_device(self, xq_ptr: int, nq: int, k: int, n_probe: int, D_ptr: int, I_ptr: int, tie_ptr: int = 0):
        _native.check(lib().vi_indexer_search_device(self._h, C.c_void_p(xq_ptr), nq, k, n_probe, C.c_void_p(D_ptr),
                                                     C.c_void_p(I_ptr), C.c_void_p(tie_ptr) if tie_ptr else None))


    def probe_device(self, xq_ptr: int, nq: int, n_probe: int, probes_ptr: int, order_ptr: int) -> int:
        """coarse step only (ivf_index.rs:205-220) -> n_probe_eff; probes / order are u32[nq][n_probe_eff] on the device"""
        p_eff = C.c_uint64(0)
        _native.check(lib().vi_indexer_probe_device(self._h, C.c_void_p(xq_ptr), nq, n_probe, C.c_void_p(probes_ptr),
                                                    C.c_void_p(order_ptr), C.byref(p_eff)))
        return int(p_eff.value)

    def search_probed_device(self, xq_ptr: int, nq: int, k: int, n_probe_eff: int, probes_ptr: int, order_ptr: int,
                             D_ptr: int, I_ptr: int, tie_ptr: int = 0):
        """list scan + top-k (ivf_index.rs:223-274) with probe lists computed elsewhere"""
        _native.check(lib().vi_indexer_search_probed_device(self._h, C.c_void_p(xq_ptr), nq, k, n_probe_eff,
                                                            C.c_void_p(probes_ptr), C.c_void_p(order_ptr),
                                                            C.c_void_p(D_ptr), C.c_void_p(I_ptr),
                                                            C.c_void_p(tie_ptr) if tie_ptr else None))


def _config(dimension, index_dir, shards_dir, **ext):
    cfg = _native.Config()
    lib().vi_config_init(C.byref(cfg), int(dimension))
    keep = (index_dir.encode(), shards_dir.encode())
    cfg.index_dir, cfg.shards_dir = keep
    for key, val in ext.items():
        if val is not None:
            setattr(cfg, key, val)
    return cfg, keep


def build(xb, work_dir: Optional[str] = None, *, nlist: int = 0, seed: int = 0, assign_mode: int = 0,
          device: int = 0, rank: int = 0, world_size: int = 0, now_secs: int = 0, ext_ids=None,
          timestamps=None) -> VectorIndex:
    """build(xb, work_dir=None) — lib.rs:220-280.  Keyword arguments are extensions."""
    xb = np.asarray(xb)
    if xb.ndim != 2:
        raise RuntimeError("Array must be 2-dimensional")
    n, d = xb.shape
    if n == 0:
        raise RuntimeError("Cannot build index from empty array")  # lib.rs:225-227
    xb = np.ascontiguousarray(xb, dtype=np.float32)
    work = work_dir if work_dir is not None else os.path.join(tempfile.gettempdir(), "vector_indexer_bench")
    index_dir, shards_dir = os.path.join(work, "index"), os.path.join(work, "shards")
    os.makedirs(index_dir, exist_ok=True)
    os.makedirs(shards_dir, exist_ok=True)
    cfg, keep = _config(d, index_dir, shards_dir, nlist_override=nlist, seed=seed, assign_mode=assign_mode,
                        device=device, rank=rank, world_size=world_size, now_secs=now_secs)
    h = C.c_void_p()
    _native.check(lib().vi_indexer_new(C.byref(cfg), C.byref(h)))
    e = None if ext_ids is None else np.ascontiguousarray(ext_ids, dtype=np.uint64)
    t = None if timestamps is None else np.ascontiguousarray(timestamps, dtype=np.uint64)
    try:
        _native.check(lib().vi_indexer_build_from_records(h, _native.ptr(e), _native.ptr(xb), _native.ptr(t), None, n),
                      prefix="Failed to build index: ")
    except Exception:
        lib().vi_indexer_free(h)
        raise
    return VectorIndex(h, d, keep)


def load(index_dir: str, shards_dir: str, dimension: int, *, device: int = 0, rank: int = 0,
         world_size: int = 0, placement: int = 0) -> VectorIndex:
    """load(index_dir, shards_dir, dimension) — lib.rs:291-304."""
    cfg, keep = _config(dimension, index_dir, shards_dir, device=device, rank=rank, world_size=world_size,
                        placement=placement)
    h = C.c_void_p()
    _native.check(lib().vi_indexer_load(C.byref(cfg), C.byref(h)), prefix="Failed to load index: ")
    return VectorIndex(h, dimension, keep)


# ---- k-means seams (src/kmeans.rs public functions) --------------------------------------------
def _kmeans(fn, X, k, max_iters, thr, seed, mode):
    X = np.asarray(X, dtype=np.float32)
    if X.ndim != 2:
        X = X.reshape(0, 0)
    X = np.ascontiguousarray(X)
    n, d = X.shape
    cent = np.zeros((int(k), d), dtype=np.float32)
    labels = np.zeros(n, dtype=np.uint64)
    iters = C.c_uint64(0)
    _native.check(fn(_native.ptr(X), n, d, int(k), int(max_iters), -1.0 if thr is None else float(thr), int(seed),
                     int(mode), _native.ptr(cent), _native.ptr(labels), C.byref(iters)))
    return cent, labels, iters.value


def kmeans_mini_batch(X, k, max_iters, early_stop_threshold=None, seed=42, mode=VI_ASSIGN_REFERENCE):
    """run_kmeans_mini_batch — src/kmeans.rs:64-70 -> (centroids, labels, iterations_run)"""
    return _kmeans(lib().vi_kmeans_mini_batch, X, k, max_iters, early_stop_threshold, seed, mode)


def kmeans_parallel(X, k, max_iters, early_stop_threshold=None, seed=42, mode=VI_ASSIGN_REFERENCE):
    """run_kmeans_parallel — src/kmeans.rs:15-21"""
    return _kmeans(lib().vi_kmeans_parallel, X, k, max_iters, early_stop_threshold, seed, mode)


def assign(X, centroids, seed=42, mode=VI_ASSIGN_REFERENCE, return_dist=False):
    """assign_points_simd_parallel — src/kmeans.rs:445-459"""
    X = np.ascontiguousarray(X, dtype=np.float32)
    cent = np.ascontiguousarray(centroids, dtype=np.float32)
    n, d = X.shape
    labels = np.zeros(n, dtype=np.uint64)
    dist = np.zeros(n, dtype=np.float32) if return_dist else None
    _native.check(lib().vi_assign(_native.ptr(X), n, d, _native.ptr(cent), cent.shape[0], int(seed), int(mode),
                                  _native.ptr(labels), _native.ptr(dist)))
    return (labels, dist) if return_dist else labels


def l2sq_pairs(a, b, order=VI_ORDER_SCALAR):
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    n, d = a.shape
    out = np.zeros(n, dtype=np.float32)
    _native.check(lib().vi_l2sq_pairs(_native.ptr(a), _native.ptr(b), n, d, int(order), _native.ptr(out)))
    return out
