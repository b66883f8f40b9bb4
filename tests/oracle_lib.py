"""ctypes loader for the CPU oracle (oracle/libvi_oracle.so).

Test infrastructure only: tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg are the only users.  The product never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_ORACLE_DIR = os.path.join(_ROOT, "oracle")
_LIB_PATH = os.path.join(_ORACLE_DIR, "libvi_oracle.so")

ORC_OK, ORC_INVALID_INPUT, ORC_NOT_FOUND, ORC_INVALID_DATA, ORC_OTHER, ORC_IO, ORC_PANIC = range(7)

u64, u32, i64, f32 = C.c_uint64, C.c_uint32, C.c_int64, C.c_float
P = C.POINTER


def usable_cpus():
    """CPUs this process may really use: affinity mask capped by the cgroup quota (the GPU box
    exposes every host core to nproc but gives the job a 16-CPU share)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, 16))


# OpenMP would otherwise start one spinning thread per visible host core
os.environ.setdefault("OMP_NUM_THREADS", str(usable_cpus()))
os.environ.setdefault("OMP_WAIT_POLICY", "passive")


def build_oracle(force=False):
    src = os.path.join(_ORACLE_DIR, "vi_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _ORACLE_DIR, "libvi_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


class OrcRng(C.Structure):
    _fields_ = [("key", u32 * 8), ("counter", u64), ("buf", u32 * 64), ("index", u32)]


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build_oracle()
    L = C.CDLL(_LIB_PATH)
    vp = C.c_void_p
    sigs = {
        "orc_calculate_num_clusters": (u64, [u64]),
        "orc_calculate_max_iterations": (u64, [u64]),
        "orc_minibatch_size": (u64, [u64]),
        "orc_meta_k": (u64, [u64]),
        "orc_num_shards": (u64, [u64]),
        "orc_l2sq_scalar": (f32, [vp, vp, C.c_size_t]),
        "orc_l2sq_simd": (f32, [vp, vp, C.c_size_t]),
        "orc_rng_seed_from_u64": (None, [P(OrcRng), u64]),
        "orc_rng_from_seed": (None, [P(OrcRng), vp]),
        "orc_rng_next_u32": (u32, [P(OrcRng)]),
        "orc_rng_next_u64": (u64, [P(OrcRng)]),
        "orc_rng_gen_range_usize": (u64, [P(OrcRng), u64, u64]),
        "orc_rng_shuffle_u64": (None, [P(OrcRng), vp, u64]),
        "orc_chacha_block": (None, [vp, u64, u64, C.c_int, vp]),
        "orc_find_nearest_centroid": (None, [vp, vp, C.c_size_t, C.c_size_t, P(u64), P(f32)]),
        "orc_assign_brute_force": (None, [vp, C.c_size_t, C.c_size_t, vp, C.c_size_t, vp]),
        "orc_assign_hierarchical": (None, [vp, C.c_size_t, C.c_size_t, vp, C.c_size_t, u64, vp]),
        "orc_assign": (None, [vp, C.c_size_t, C.c_size_t, vp, C.c_size_t, u64, vp]),
        "orc_build_centroid_hierarchy": (None, [vp, C.c_size_t, C.c_size_t, C.c_size_t, u64, vp, vp]),
        "orc_kmeans_pp_init": (None, [vp, C.c_size_t, C.c_size_t, C.c_size_t, u64, vp]),
        "orc_update_centroids": (None, [vp, C.c_size_t, C.c_size_t, vp, C.c_size_t, vp, vp]),
        "orc_kmeans_mini_batch": (C.c_int, [vp, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, f32, u64, C.c_int, vp, vp, P(u64)]),
        "orc_kmeans_parallel": (C.c_int, [vp, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, f32, u64, C.c_int, vp, vp, P(u64)]),
        "orc_shard_save_to": (C.c_int, [C.c_char_p, u64, u32, u32, vp, vp, vp, vp, vp, vp, vp]),
        "orc_shard_get_centroid_vectors_from": (C.c_int, [C.c_char_p, u64, vp, C.c_size_t, P(u32), vp, vp, vp, vp]),
        "orc_index_build": (C.c_int, [vp, vp, vp, C.c_size_t, u32, u64, u64, u64, C.c_int, C.c_char_p, C.c_char_p, P(vp)]),
        "orc_index_load": (C.c_int, [C.c_char_p, C.c_char_p, P(vp)]),
        "orc_index_free": (None, [vp]),
        "orc_index_num_centroids": (u64, [vp]),
        "orc_index_dimension": (u32, [vp]),
        "orc_index_num_shards": (u64, [vp]),
        "orc_index_centroids": (None, [vp, vp, vp]),
        "orc_index_list_len": (u64, [vp, u64]),
        "orc_index_search": (C.c_int, [vp, vp, u64, u64, vp, vp, vp, P(u64)]),
        "orc_index_search_batch": (C.c_int, [vp, vp, u64, u64, u64, C.c_int, vp, vp]),
        "orc_index_probe": (C.c_int, [vp, vp, u64, vp, P(u64)]),
        "orc_index_search_partial": (C.c_int, [vp, vp, u64, u64, u32, u32, vp, vp, vp, P(u64)]),
        "orc_max_threads": (C.c_int, []),
    }
    for name, (res, args) in sigs.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def f32c(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def l2sq_scalar(a, b):
    a, b = f32c(a), f32c(b)
    return np.float32(lib().orc_l2sq_scalar(_p(a), _p(b), a.size))


def l2sq_simd(a, b):
    a, b = f32c(a), f32c(b)
    return np.float32(lib().orc_l2sq_simd(_p(a), _p(b), a.size))


def assign(X, Cn, seed=42, mode="auto"):
    X, Cn = f32c(X), f32c(Cn)
    n, d = X.shape
    k = Cn.shape[0]
    labels = np.zeros(n, dtype=np.uint64)
    L = lib()
    if mode == "brute":
        L.orc_assign_brute_force(_p(X), n, d, _p(Cn), k, _p(labels))
    elif mode == "hier":
        L.orc_assign_hierarchical(_p(X), n, d, _p(Cn), k, seed, _p(labels))
    else:
        L.orc_assign(_p(X), n, d, _p(Cn), k, seed, _p(labels))
    return labels


def build_centroid_hierarchy(Cn, meta_k, seed):
    Cn = f32c(Cn)
    k, d = Cn.shape
    meta = np.zeros((meta_k, d), dtype=np.float32)
    c2m = np.zeros(k, dtype=np.uint64)
    lib().orc_build_centroid_hierarchy(_p(Cn), k, d, meta_k, seed, _p(meta), _p(c2m))
    return meta, c2m


def kmeans_pp_init(X, k, seed):
    X = f32c(X)
    n, d = X.shape
    Cn = np.zeros((k, d), dtype=np.float32)
    lib().orc_kmeans_pp_init(_p(X), n, d, k, seed, _p(Cn))
    return Cn


def update_centroids(X, labels, k):
    X = f32c(X)
    n, d = X.shape
    labels = np.ascontiguousarray(labels, dtype=np.uint64)
    Cn = np.zeros((k, d), dtype=np.float32)
    counts = np.zeros(k, dtype=np.uint64)
    lib().orc_update_centroids(_p(X), n, d, _p(labels), k, _p(Cn), _p(counts))
    return Cn, counts


def _kmeans(fn, X, k, max_iters, thr, seed, force_brute, want_labels=True):
    X = f32c(X)
    if X.ndim != 2:
        X = X.reshape(0, 0)
    n, d = X.shape
    Cn = np.zeros((k, max(d, 1)), dtype=np.float32)[:, :d].copy()
    labels = np.zeros(n, dtype=np.uint64) if want_labels else None
    iters = u64(0)
    rc = fn(_p(X), n, d, k, max_iters, -1.0 if thr is None else thr, seed, int(force_brute),
            _p(Cn), _p(labels), C.byref(iters))
    return rc, Cn, labels, iters.value


def kmeans_mini_batch(X, k, max_iters, thr=None, seed=42, force_brute=False, want_labels=True):
    """want_labels=False: the training loop only (labels None) — for inputs whose final assignment the test checks on
    sampled rows"""
    return _kmeans(lib().orc_kmeans_mini_batch, X, k, max_iters, thr, seed, force_brute, want_labels)


def kmeans_parallel(X, k, max_iters, thr=None, seed=42, force_brute=False):
    return _kmeans(lib().orc_kmeans_parallel, X, k, max_iters, thr, seed, force_brute)


def shard_save(shards_dir, shard_id, dim, centroid_ids, centroid_vecs, lists):
    """lists: list of (ids, ext_ids, timestamps, vecs[n x dim])"""
    nl = len(lists)
    off = np.zeros(nl + 1, dtype=np.uint64)
    for i, l in enumerate(lists):
        off[i + 1] = off[i] + len(l[0])
    tot = int(off[-1])
    ids = np.zeros(tot, dtype=np.uint64)
    eids = np.zeros(tot, dtype=np.uint64)
    tss = np.zeros(tot, dtype=np.uint64)
    vecs = np.zeros((tot, dim), dtype=np.float32)
    for i, l in enumerate(lists):
        a, b = int(off[i]), int(off[i + 1])
        if b > a:
            ids[a:b], eids[a:b], tss[a:b] = l[0], l[1], l[2]
            vecs[a:b] = np.asarray(l[3], dtype=np.float32).reshape(b - a, dim)
    cids = np.ascontiguousarray(centroid_ids, dtype=np.uint64)
    cv = f32c(np.asarray(centroid_vecs, dtype=np.float32).reshape(nl, dim))
    return lib().orc_shard_save_to(shards_dir.encode(), shard_id, dim, nl, _p(cids), _p(cv), _p(off),
                                   _p(ids), _p(eids), _p(tss), _p(vecs))


def shard_get(shards_dir, shard_id, centroid_ids):
    cids = np.ascontiguousarray(centroid_ids, dtype=np.uint64)
    n = len(cids)
    counts = np.zeros(n, dtype=np.uint64)
    dim = u32(0)
    L = lib()
    rc = L.orc_shard_get_centroid_vectors_from(shards_dir.encode(), shard_id, _p(cids), n, C.byref(dim),
                                               _p(counts), None, None, None)
    if rc != ORC_OK:
        return rc, None
    tot = int(counts.sum())
    cent = np.zeros((n, dim.value), dtype=np.float32)
    metas = np.zeros((tot, 3), dtype=np.uint64)
    vecs = np.zeros((tot, dim.value), dtype=np.float32)
    rc = L.orc_shard_get_centroid_vectors_from(shards_dir.encode(), shard_id, _p(cids), n, C.byref(dim),
                                               _p(counts), _p(cent), _p(metas), _p(vecs))
    out, o = [], 0
    for i in range(n):
        c = int(counts[i])
        out.append((int(cids[i]), cent[i], metas[o:o + c], vecs[o:o + c]))
        o += c
    return rc, out


class OracleIndex:
    def __init__(self, handle):
        self.h = handle

    @staticmethod
    def build(X, index_dir, shards_dir, ext_ids=None, timestamps=None, nlist=0, seed=42, now=1_700_000_000,
              force_brute=False):
        X = f32c(X)
        n, d = X.shape
        e = None if ext_ids is None else np.ascontiguousarray(ext_ids, dtype=np.uint64)
        t = None if timestamps is None else np.ascontiguousarray(timestamps, dtype=np.uint64)
        h = C.c_void_p()
        rc = lib().orc_index_build(_p(X), _p(e), _p(t), n, d, nlist, seed, now, int(force_brute),
                                   index_dir.encode(), shards_dir.encode(), C.byref(h))
        if rc != ORC_OK:
            raise RuntimeError(f"orc_index_build rc={rc}")
        return OracleIndex(h)

    @staticmethod
    def load(index_dir, shards_dir):
        h = C.c_void_p()
        rc = lib().orc_index_load(index_dir.encode(), shards_dir.encode(), C.byref(h))
        if rc != ORC_OK:
            raise RuntimeError(f"orc_index_load rc={rc}")
        return OracleIndex(h)

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_index_free(self.h)
            self.h = None

    @property
    def num_centroids(self):
        return lib().orc_index_num_centroids(self.h)

    @property
    def dimension(self):
        return lib().orc_index_dimension(self.h)

    @property
    def num_shards(self):
        return lib().orc_index_num_shards(self.h)

    def list_len(self, c):
        return int(lib().orc_index_list_len(self.h, c))

    def centroids(self):
        k, d = self.num_centroids, self.dimension
        Cn = np.zeros((k, d), dtype=np.float32)
        c2s = np.zeros(k, dtype=np.uint64)
        lib().orc_index_centroids(self.h, _p(Cn), _p(c2s))
        return Cn, c2s

    def list_len(self, c):
        return lib().orc_index_list_len(self.h, c)

    def search(self, q, k, n_probe, include_vectors=False):
        q = f32c(q)
        ids = np.zeros(max(k, 1), dtype=np.uint64)
        ds = np.zeros(max(k, 1), dtype=np.float32)
        vs = np.zeros((max(k, 1), self.dimension), dtype=np.float32) if include_vectors else None
        cnt = u64(0)
        rc = lib().orc_index_search(self.h, _p(q), k, n_probe, _p(ids), _p(ds), _p(vs), C.byref(cnt))
        c = cnt.value
        return rc, ids[:c], ds[:c], (vs[:c] if include_vectors else None)

    def search_batch(self, Q, k, n_probe, threads=0):
        Q = f32c(Q)
        nq = Q.shape[0]
        D = np.zeros((nq, k), dtype=np.float32)
        I = np.zeros((nq, k), dtype=np.int64)
        rc = lib().orc_index_search_batch(self.h, _p(Q), nq, k, n_probe, threads, _p(D), _p(I))
        return rc, D, I

    def search_partial_batch(self, Q, k, n_probe, rank, world):
        """per-rank (D, I, tie) padded with +inf / -1 / 2^64-1 (multi-GPU protocol checker)"""
        Q = f32c(Q)
        nq = Q.shape[0]
        D = np.full((nq, k), np.inf, dtype=np.float32)
        I = np.full((nq, k), -1, dtype=np.int64)
        T = np.full((nq, k), np.iinfo(np.uint64).max, dtype=np.uint64)
        ids = np.zeros(k, dtype=np.uint64)
        ds = np.zeros(k, dtype=np.float32)
        ts = np.zeros(k, dtype=np.uint64)
        for i in range(nq):
            cnt = u64(0)
            rc = lib().orc_index_search_partial(self.h, _p(Q[i]), k, n_probe, rank, world, _p(ids), _p(ds), _p(ts),
                                                C.byref(cnt))
            assert rc == ORC_OK
            c = cnt.value
            D[i, :c], I[i, :c], T[i, :c] = ds[:c], ids[:c].astype(np.int64), ts[:c]
        return D, I, T

    def probe(self, q, n_probe):
        q = f32c(q)
        out = np.zeros(max(n_probe, 1), dtype=np.uint64)
        cnt = u64(0)
        rc = lib().orc_index_probe(self.h, _p(q), n_probe, _p(out), C.byref(cnt))
        return rc, out[:cnt.value]
