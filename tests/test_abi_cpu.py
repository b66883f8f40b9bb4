"""CPU: the C-ABI library loads, exports everything include/vi_amd.h declares, and its host-side
code (heuristics, shard writer/reader, index.bin codec, error kinds) matches the golden fixtures.
No GPU compute is called here."""
import ctypes as C
import json
import os
import re

import numpy as np
import pytest

import vector_indexer_py as vip
from vector_indexer_py import _native as N

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")


def load(name):
    with open(os.path.join(G, name)) as f:
        return json.load(f)


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "vi_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b(vi_[a-z0-9_]+)\s*\(", hdr))
    assert len(names) >= 25
    L = C.CDLL(N.LIB_PATH)
    for n in sorted(names):
        assert hasattr(L, n), f"{n} declared in include/vi_amd.h but not exported"
    assert set(N.SIGNATURES) == names
    assert N.lib().vi_abi_version() == 2


def test_heuristics_match_golden():
    L = N.lib()
    for r in load("heuristics.json"):
        assert L.vi_calculate_num_clusters(r["n"]) == r["k"]
        assert L.vi_calculate_max_iterations(r["n"]) == r["max_iters"]
        assert L.vi_minibatch_size(r["n"]) == r["batch"]
        assert vip.suggest_nlist(r["n"]) == r["k"]


def _save(tmp, sc):
    nl = len(sc["lists"])
    off = np.zeros(nl + 1, dtype=np.uint64)
    for i, l in enumerate(sc["lists"]):
        off[i + 1] = off[i] + len(l["vectors"])
    tot = int(off[-1])
    flat = [v for l in sc["lists"] for v in l["vectors"]]
    ids = np.array([v["id"] for v in flat], dtype=np.uint64).reshape(tot)
    ext = np.array([v["ext"] for v in flat], dtype=np.uint64).reshape(tot)
    ts = np.array([v["ts"] for v in flat], dtype=np.uint64).reshape(tot)
    vecs = np.array([v["v"] for v in flat], dtype=np.float32).reshape(tot, sc["dim"])
    cids = np.array([l["centroid_id"] for l in sc["lists"]], dtype=np.uint64)
    cv = np.array([l["centroid"] for l in sc["lists"]], dtype=np.float32).reshape(nl, sc["dim"])
    return N.lib().vi_shard_save_to(str(tmp).encode(), sc["shard_id"], sc["dim"], nl, N.ptr(cids), N.ptr(cv),
                                    N.ptr(off), N.ptr(ids), N.ptr(ext), N.ptr(ts), N.ptr(vecs))


def _get(tmp, shard_id, cids):
    cids = np.ascontiguousarray(cids, dtype=np.uint64)
    counts = np.zeros(len(cids), dtype=np.uint64)
    dim = C.c_uint32(0)
    L = N.lib()
    rc = L.vi_shard_get_centroid_vectors_from(str(tmp).encode(), shard_id, N.ptr(cids), len(cids), C.byref(dim),
                                              N.ptr(counts), None, None, None)
    if rc != 0:
        return rc, None
    tot = int(counts.sum())
    cent = np.zeros((len(cids), dim.value), dtype=np.float32)
    metas = np.zeros((tot, 3), dtype=np.uint64)
    vecs = np.zeros((tot, dim.value), dtype=np.float32)
    rc = L.vi_shard_get_centroid_vectors_from(str(tmp).encode(), shard_id, N.ptr(cids), len(cids), C.byref(dim),
                                              N.ptr(counts), N.ptr(cent), N.ptr(metas), N.ptr(vecs))
    return rc, (counts, cent, metas, vecs)


@pytest.mark.parametrize("sc", load("shards.json"), ids=lambda s: s["name"])
def test_shard_writer_is_byte_identical(sc, tmp_path):
    """Shard::save_to layout (src/shards.rs:68-177) — bytes derived from the repr(C) structs."""
    assert _save(tmp_path, sc) == 0
    raw = open(tmp_path / f"shard_{sc['shard_id']}.bin", "rb").read()
    assert raw.hex() == sc["hex"]
    cids = [l["centroid_id"] for l in sc["lists"]]
    rc, got = _get(tmp_path, sc["shard_id"], list(reversed(cids)))
    assert rc == 0
    counts, cent, metas, vecs = got
    exp = list(reversed(sc["lists"]))
    assert counts.tolist() == [len(l["vectors"]) for l in exp]
    assert cent.tobytes() == np.array([l["centroid"] for l in exp], dtype=np.float32).tobytes()
    flat = [v for l in exp for v in l["vectors"]]
    assert metas.tolist() == [[v["id"], v["ext"], v["ts"]] for v in flat]
    assert vecs.tobytes() == np.array([v["v"] for v in flat], dtype=np.float32).tobytes()


def test_shard_reader_error_kinds(tmp_path):
    """tests/shards_tests.rs:541-630: missing file, unknown centroid, corrupt header."""
    sc = load("shards.json")[0]
    rc, _ = _get(tmp_path, 999, [1])
    assert rc == N.VI_ERR_OTHER
    assert _save(tmp_path, sc) == 0
    rc, _ = _get(tmp_path, sc["shard_id"], [12345])
    assert rc == N.VI_ERR_NOT_FOUND
    assert b"Centroid 12345 not found" in N.lib().vi_last_error()
    p = tmp_path / f"shard_{sc['shard_id']}.bin"
    raw = bytearray(p.read_bytes())
    raw[0:4] = b"\xff\xff\xff\xff"
    p.write_bytes(bytes(raw))
    rc, _ = _get(tmp_path, sc["shard_id"], [5])
    assert rc == N.VI_ERR_INVALID_DATA
    # overwrite on re-save (:672-709)
    assert _save(tmp_path, sc) == 0
    rc, _ = _get(tmp_path, sc["shard_id"], [5])
    assert rc == 0


def test_load_missing_index_is_an_error(tmp_path):
    """tests/api_tests.rs: load of a missing index -> Err (raw fs NotFound)."""
    with pytest.raises(RuntimeError) as e:
        vip.load(str(tmp_path / "nope"), str(tmp_path / "nope_s"), 8)
    assert e.value.status == N.VI_ERR_NOT_FOUND


def test_build_argument_errors_without_gpu(tmp_path):
    """Validation happens before any device work (api.rs:116-134)."""
    L = N.lib()
    cfg = N.Config()
    L.vi_config_init(C.byref(cfg), 8)
    assert (cfg.default_k, cfg.default_n_probe, cfg.max_k, cfg.max_n_probe) == (10, 20, 10000, 10000)  # api.rs:38-41
    keep = (str(tmp_path / "i").encode(), str(tmp_path / "s").encode())
    cfg.index_dir, cfg.shards_dir = keep
    h = C.c_void_p()
    assert L.vi_indexer_new(C.byref(cfg), C.byref(h)) == 0
    x = np.zeros((3, 8), dtype=np.float32)
    assert L.vi_indexer_build_from_records(h, None, N.ptr(x), None, None, 0) == N.VI_ERR_INVALID_INPUT
    assert L.vi_last_error() == b"no vectors provided"
    dims = np.array([8, 7, 8], dtype=np.uint32)
    assert L.vi_indexer_build_from_records(h, None, N.ptr(x), None, N.ptr(dims), 3) == N.VI_ERR_INVALID_INPUT
    assert L.vi_last_error() == b"vector dimension mismatch at index 1: expected 8, got 7"
    # search argument validation (api.rs:189-201, ivf_index.rs:197-202)
    D = np.zeros((1, 4), dtype=np.float32)
    I = np.zeros((1, 4), dtype=np.int64)
    q = np.zeros((1, 8), dtype=np.float32)
    assert L.vi_indexer_search(h, N.ptr(q), 1, 7, 4, 4, N.ptr(D), N.ptr(I), None, None, None) == N.VI_ERR_INVALID_INPUT
    assert b"query dimension mismatch: expected 8, got 7" == L.vi_last_error()
    assert L.vi_indexer_search(h, N.ptr(q), 1, 8, 0, 4, N.ptr(D), N.ptr(I), None, None, None) == N.VI_ERR_INVALID_INPUT
    assert L.vi_indexer_search(h, N.ptr(q), 1, 8, 4, 0, N.ptr(D), N.ptr(I), None, None, None) == N.VI_ERR_INVALID_INPUT
    assert L.vi_last_error() == b"k and n_probe must be greater than 0"
    L.vi_indexer_free(h)


def test_no_cpu_fallback_without_device(tmp_path):
    """On a box without a GPU every compute entry point fails loudly with VI_ERR_DEVICE."""
    if N.lib().vi_device_count() > 0:
        pytest.skip("GPU present")
    a = np.zeros((2, 4), dtype=np.float32)
    with pytest.raises(RuntimeError) as e:
        vip.l2sq_pairs(a, a)
    assert e.value.status == N.VI_ERR_DEVICE
    with pytest.raises(RuntimeError) as e:
        vip.build(np.zeros((20, 4), dtype=np.float32), str(tmp_path))
    assert e.value.status in (N.VI_ERR_DEVICE, N.VI_ERR_PANIC, N.VI_ERR_OTHER)
