"""ctypes declarations for libvi_amd.so (include/vi_amd.h).  No torch, no CPU fallback."""
import ctypes as C
import os

VI_OK, VI_ERR_INVALID_INPUT, VI_ERR_NOT_FOUND, VI_ERR_INVALID_DATA, VI_ERR_OTHER, VI_ERR_IO, VI_ERR_PANIC, \
    VI_ERR_DEVICE = range(8)
VI_ORDER_SCALAR, VI_ORDER_LANES = 0, 1
VI_ASSIGN_REFERENCE, VI_ASSIGN_EXACT = 0, 1

_STATUS_NAME = {1: "InvalidInput", 2: "NotFound", 3: "InvalidData", 4: "Other", 5: "Io", 6: "Panic", 7: "Device"}

LIB_PATH = os.environ.get("VI_AMD_LIB") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                      "libvi_amd.so")

u64, u32, i32, i64, f32 = C.c_uint64, C.c_uint32, C.c_int32, C.c_int64, C.c_float
vp = C.c_void_p


class ViError(RuntimeError):
    """RuntimeError carrying the io::ErrorKind-like status of the failed call."""

    def __init__(self, status, message):
        super().__init__(message)
        self.status = status
        self.kind = _STATUS_NAME.get(status, str(status))


class Config(C.Structure):
    _fields_ = [("dimension", u32), ("index_dir", C.c_char_p), ("shards_dir", C.c_char_p),
                ("default_k", u64), ("default_n_probe", u64), ("max_k", u64), ("max_n_probe", u64),
                ("nlist_override", u64), ("seed", u64), ("assign_mode", i32), ("device", i32),
                ("rank", i32), ("world_size", i32), ("now_secs", u64), ("placement", i32), ("reserved0", i32)]


class SearchStats(C.Structure):
    _fields_ = [("nq", u64), ("k", u64), ("n_probe_eff", u64), ("coarse_candidates", u64),
                ("scanned_vectors", u64), ("scan_items", u64), ("ms_total", f32), ("ms_coarse", f32),
                ("ms_group", f32), ("ms_scan", f32), ("ms_merge", f32), ("fallback_queries", u64),
                ("filter_tile_blocks", u64), ("filter_rechecked", u64), ("filter_accepted", u64),
                ("rank_mode", u64), ("group_queries", u64)]


class BuildStats(C.Structure):
    _fields_ = [("n", u64), ("nlist", u64), ("lists", u64), ("shards", u64), ("shard_bytes", u64), ("ms_total", f32),
                ("ms_upload", f32), ("ms_kmeans", f32), ("ms_group", f32), ("ms_super", f32), ("ms_export", f32),
                ("ms_index", f32)]


FETCH_ROWS_FN = C.CFUNCTYPE(C.c_int, vp, C.POINTER(u64), u64, vp)


class RowSource(C.Structure):
    """vi_row_source: fetch_rows(ctx, global_rows, n_rows, out_dev) -> 0 on success"""
    _fields_ = [("ctx", vp), ("fetch_rows", FETCH_ROWS_FN)]


class AssignStats(C.Structure):
    _fields_ = [("n", u64), ("k", u64), ("ambiguous_rows", u64), ("used_mfma", u32), ("ms_total", f32),
                ("ms_filter", f32), ("tier1_rows", u64), ("ambiguous_rows_dev", vp), ("ambiguous_cap", u64)]


SIGNATURES = {
    "vi_last_error": (C.c_char_p, []),
    "vi_abi_version": (u32, []),
    "vi_device_count": (C.c_int, []),
    "vi_calculate_num_clusters": (u64, [u64]),
    "vi_calculate_max_iterations": (u64, [u64]),
    "vi_minibatch_size": (u64, [u64]),
    "vi_l2sq_pairs": (C.c_int, [vp, vp, u64, u32, C.c_int, vp]),
    "vi_assign": (C.c_int, [vp, u64, u32, vp, u64, u64, C.c_int, vp, vp]),
    "vi_assign_device": (C.c_int, [i32, vp, u64, u32, vp, u64, u64, C.c_int, vp, vp]),
    "vi_kmeans_mini_batch": (C.c_int, [vp, u64, u32, u64, u64, f32, u64, C.c_int, vp, vp, C.POINTER(u64)]),
    "vi_kmeans_parallel": (C.c_int, [vp, u64, u32, u64, u64, f32, u64, C.c_int, vp, vp, C.POINTER(u64)]),
    "vi_kmeans_mini_batch_device": (C.c_int, [i32, vp, u64, u32, u64, u64, f32, u64, C.c_int, vp, vp, C.POINTER(u64)]),
    "vi_kmeans_parallel_device": (C.c_int, [i32, vp, u64, u32, u64, u64, f32, u64, C.c_int, vp, vp, C.POINTER(u64)]),
    "vi_kmeans_mini_batch_train": (C.c_int, [i32, vp, u64, u32, u64, u64, f32, u64, vp, C.POINTER(u64)]),
    "vi_kmeans_pp_init": (C.c_int, [i32, vp, u64, u32, u64, u64, vp]),
    "vi_kmeans_partial_sums_device": (C.c_int, [i32, vp, u64, u32, vp, u64, vp, vp]),
    "vi_kmeans_finish_update_device": (C.c_int, [i32, vp, vp, u64, u32, vp, vp, C.POINTER(f32), vp, C.POINTER(u64)]),
    "vi_kmeans_centroid_delta_device": (C.c_int, [i32, vp, vp, u64, u32, C.POINTER(f32)]),
    "vi_rng_seed_from_u64": (vp, [u64]),
    "vi_rng_gen_range": (u64, [vp, u64, u64]),
    "vi_rng_free": (None, [vp]),
    "vi_shard_save_to": (C.c_int, [C.c_char_p, u64, u32, u32, vp, vp, vp, vp, vp, vp, vp]),
    "vi_shard_get_centroid_vectors_from": (C.c_int, [C.c_char_p, u64, vp, u64, C.POINTER(u32), vp, vp, vp, vp]),
    "vi_config_init": (None, [C.POINTER(Config), u32]),
    "vi_indexer_new": (C.c_int, [C.POINTER(Config), C.POINTER(vp)]),
    "vi_indexer_load": (C.c_int, [C.POINTER(Config), C.POINTER(vp)]),
    "vi_indexer_build_from_records": (C.c_int, [vp, vp, vp, vp, vp, u64]),
    "vi_indexer_build_from_vector_file": (C.c_int, [vp, C.c_char_p]),
    "vi_indexer_search": (C.c_int, [vp, vp, u64, u32, u64, u64, vp, vp, vp, vp, C.POINTER(u64)]),
    "vi_indexer_search_device": (C.c_int, [vp, vp, u64, u64, u64, vp, vp, vp]),
    "vi_indexer_probe_device": (C.c_int, [vp, vp, u64, u64, vp, vp, C.POINTER(u64)]),
    "vi_indexer_search_probed_device": (C.c_int, [vp, vp, u64, u64, u64, vp, vp, vp, vp, vp]),
    "vi_merge_partials_device": (C.c_int, [i32, u64, u64, u32, vp, vp, vp, vp, vp]),
    "vi_packed_result_bytes": (u64, [u64, u64]),
    "vi_merge_partials_packed_device": (C.c_int, [i32, u64, u64, u32, vp, vp, vp]),
    "vi_indexer_dimension": (u32, [vp]),
    "vi_indexer_num_centroids": (u64, [vp]),
    "vi_indexer_num_vectors": (u64, [vp]),
    "vi_indexer_num_shards": (u64, [vp]),
    "vi_indexer_centroids": (C.c_int, [vp, vp, vp]),
    "vi_indexer_free": (None, [vp]),
    "vi_indexer_last_stats": (C.c_int, [vp, C.POINTER(SearchStats)]),
    "vi_indexer_enable_timing": (None, [vp, C.c_int]),
    "vi_indexer_last_build_stats": (C.c_int, [vp, C.POINTER(BuildStats)]),
}

_lib = None


def lib():
    """Load libvi_amd.so.  Raises (never falls back) when the HIP library is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ViError(VI_ERR_DEVICE, f"{LIB_PATH} not found: run `python -c 'import __graft_entry__ as g; g.build()'`")
    L = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def check(status, prefix=""):
    if status != VI_OK:
        msg = lib().vi_last_error()
        raise ViError(status, prefix + (msg.decode("utf-8", "replace") if msg else f"status {status}"))
