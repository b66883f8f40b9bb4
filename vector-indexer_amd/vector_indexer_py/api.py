"""Mirror of the reference's public Rust API (src/api.rs) on top of the C ABI.

    VectorIndexerConfig.new(dimension).with_index_dir(..).with_shards_dir(..)      api.rs:32-54
    VectorRecord(external_id, values, timestamp=None)                              api.rs:57-62
    SearchRequest(query, include_vectors, k, n_probe) + with_*                     api.rs:64-87
    SearchResult(external_id, distance, vector)                                    api.rs:89-94
    VectorIndexer.new(cfg) / .load(cfg) / .build_from_records / .build_from_vector_file /
    .search(req) / .search_request(query) / .config()                             api.rs:101-237

Errors are ViError (a RuntimeError) whose .kind is the io::ErrorKind name the reference returns.
"""
import ctypes as C
from dataclasses import dataclass, field, replace
from typing import List, Optional

import numpy as np

from . import _native
from ._native import ViError, lib


@dataclass
class VectorIndexerConfig:
    dimension: int
    index_dir: str = "index"
    shards_dir: str = "shards"
    default_k: int = 10
    default_n_probe: int = 20
    max_k: int = 10_000
    max_n_probe: int = 10_000
    # extensions (zero = reference behaviour)
    nlist_override: int = 0
    seed: int = 0
    assign_mode: int = 0
    device: int = 0
    rank: int = 0
    world_size: int = 0
    now_secs: int = 0

    @staticmethod
    def new(dimension: int) -> "VectorIndexerConfig":
        return VectorIndexerConfig(dimension)

    def with_index_dir(self, index_dir) -> "VectorIndexerConfig":
        return replace(self, index_dir=str(index_dir))

    def with_shards_dir(self, shards_dir) -> "VectorIndexerConfig":
        return replace(self, shards_dir=str(shards_dir))


@dataclass
class VectorRecord:
    external_id: int
    values: List[float]
    timestamp: Optional[int] = None


@dataclass
class SearchRequest:
    query: List[float]
    include_vectors: bool = False
    k: int = 10
    n_probe: int = 20

    def with_k(self, k):
        return replace(self, k=k)

    def with_n_probe(self, n_probe):
        return replace(self, n_probe=n_probe)

    def with_include_vectors(self, include_vectors):
        return replace(self, include_vectors=include_vectors)


@dataclass
class SearchResult:
    external_id: int
    distance: float
    vector: Optional[List[float]] = field(default=None)


class VectorIndexer:
    def __init__(self, cfg: VectorIndexerConfig, handle):
        self._cfg = cfg
        self._h = handle

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            lib().vi_indexer_free(h)

    @staticmethod
    def _native_cfg(cfg: VectorIndexerConfig):
        c = _native.Config()
        lib().vi_config_init(C.byref(c), int(cfg.dimension))
        keep = (cfg.index_dir.encode(), cfg.shards_dir.encode())
        c.index_dir, c.shards_dir = keep
        for f in ("default_k", "default_n_probe", "max_k", "max_n_probe", "nlist_override", "seed", "assign_mode",
                  "device", "rank", "world_size", "now_secs"):
            setattr(c, f, int(getattr(cfg, f)))
        return c, keep

    @staticmethod
    def new(cfg: VectorIndexerConfig) -> "VectorIndexer":
        c, keep = VectorIndexer._native_cfg(cfg)
        h = C.c_void_p()
        _native.check(lib().vi_indexer_new(C.byref(c), C.byref(h)))
        return VectorIndexer(cfg, h)

    @staticmethod
    def load(cfg: VectorIndexerConfig) -> "VectorIndexer":
        c, keep = VectorIndexer._native_cfg(cfg)
        h = C.c_void_p()
        _native.check(lib().vi_indexer_load(C.byref(c), C.byref(h)))
        return VectorIndexer(cfg, h)

    def build_from_records(self, records: List[VectorRecord]) -> "VectorIndexer":
        n = len(records)
        dim = self._cfg.dimension
        dims = np.array([len(r.values) for r in records], dtype=np.uint32)
        vals = np.zeros((n, dim), dtype=np.float32)
        for i, r in enumerate(records):
            if len(r.values) == dim:
                vals[i] = np.asarray(r.values, dtype=np.float32)
        ext = np.array([r.external_id for r in records], dtype=np.uint64)
        ts = np.array([r.timestamp or 0 for r in records], dtype=np.uint64)  # unwrap_or(0), api.rs:138
        _native.check(lib().vi_indexer_build_from_records(self._h, _native.ptr(ext), _native.ptr(vals), _native.ptr(ts),
                                                          _native.ptr(dims), n))
        return self

    def build_from_vector_file(self, path) -> "VectorIndexer":
        _native.check(lib().vi_indexer_build_from_vector_file(self._h, str(path).encode()))
        return self

    def search(self, req: SearchRequest) -> List[SearchResult]:
        q = np.ascontiguousarray(np.asarray(req.query, dtype=np.float32).reshape(1, -1))
        k = max(int(req.k), 0)
        kcap = min(k, self._cfg.max_k)
        D = np.full((1, max(kcap, 1)), np.inf, dtype=np.float32)
        I = np.full((1, max(kcap, 1)), -1, dtype=np.int64)
        V = np.zeros((1, max(kcap, 1), self._cfg.dimension), dtype=np.float32) if req.include_vectors else None
        cnt = np.zeros(1, dtype=np.uint64)
        kout = C.c_uint64(0)
        _native.check(lib().vi_indexer_search(self._h, _native.ptr(q), 1, q.shape[1], k, int(req.n_probe),
                                              _native.ptr(D), _native.ptr(I), _native.ptr(V), _native.ptr(cnt),
                                              C.byref(kout)))
        return [SearchResult(int(I[0, j]), float(D[0, j]), V[0, j].tolist() if V is not None else None)
                for j in range(int(cnt[0]))]

    def search_request(self, query) -> SearchRequest:
        return SearchRequest(list(query), False, self._cfg.default_k, self._cfg.default_n_probe)

    def config(self) -> VectorIndexerConfig:
        return self._cfg
