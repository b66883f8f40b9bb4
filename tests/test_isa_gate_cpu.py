"""Build-time gate for the hand-placed waits of the double-buffered rank kernels (filter_search.hip, NBUF == 2).

Their block loop issues the next tile's LDS-DMA as inline asm and ends with a counted `s_waitcnt vmcnt(1)` that is only
correct if the wave's ONE younger vector-memory operation is its pair-record store.  A register spill (scratch traffic),
a store the compiler splits in two or an extra global load inside the loop would be a second younger operation: the
barrier would release before the DMA has landed, a silent LDS race no parity test is guaranteed to catch.  This test
compiles the file to gfx950 assembly (no GPU needed) and checks exactly that, so a compiler or source change that
breaks the assumption fails here."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "vector-indexer_amd", "csrc", "filter_search.hip")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.fixture(scope="module")
def asm(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not installed")
    out = str(tmp_path_factory.mktemp("isa") / "filter_search.s")
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize"]
    subprocess.check_call([HIPCC, *flags, "--cuda-device-only", "-S", "-o", out, SRC], stderr=subprocess.DEVNULL)
    return open(out).read()


def kernels(asm_text):
    """name -> (body text, metadata dict) of every filter_kernel instantiation"""
    out = {}
    for m in re.finditer(r"^(_ZN2vi12_GLOBAL__N_113filter_kernelILi(\d+)ELi(\d)ELb([01])ELi(\d)ELi(\d+)EEEvNS0_10FilterArgsE):[^\n]*\n(.*?)\n\s*s_endpgm",
                         asm_text, re.S | re.M):
        name, ng, nbuf, table, rank, gq, body = m.groups()
        meta = re.search(r"\.amdhsa_kernel " + re.escape(name) + r"\n(.*?)\.end_amdhsa_kernel", asm_text, re.S).group(1)
        out[name] = dict(body=body, meta=meta, ng=int(ng), nbuf=int(nbuf), table=table == "1", rank=int(rank), gq=int(gq))
    return out


def block_loop(body):
    """the instructions of the loop that holds the per-block s_barrier: LLVM marks a loop's header with
    '=>This Inner Loop Header' / '=>This Loop Header' and its other blocks with 'in Loop: Header=<label>'"""
    blocks, cur, label, note = {}, [], None, ""
    for line in body.split("\n"):
        m = re.match(r"^(\.LBB\d+_\d+):\s*(;.*)?$", line)
        if m:
            if label is not None:
                blocks[label] = (note, "\n".join(cur))
            label, note, cur = m.group(1), m.group(2) or "", []
        elif label is not None:
            cur.append(line)
    if label is not None:
        blocks[label] = (note, "\n".join(cur))
    groups = {}
    for lab, (note, text) in blocks.items():
        h = re.search(r"in Loop: Header=(BB\d+_\d+)", note)
        if h:
            groups.setdefault(".L" + h.group(1), []).append(text)
        elif "Loop Header" in note:
            groups.setdefault(lab, []).append(text)
    with_barrier = [t for t in groups.values() if any("s_barrier" in x for x in t)]
    assert len(with_barrier) == 1, "expected exactly one loop with a barrier"
    return "\n".join(with_barrier[0])


def test_rank_kernels_do_not_spill_and_keep_one_store_per_block(asm):
    ks = kernels(asm)
    assert len(ks) >= 16
    dbl = {n: k for n, k in ks.items() if k["nbuf"] >= 2}
    assert dbl, "no double-buffered instantiation found"
    for name, k in ks.items():
        priv = int(re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", k["meta"]).group(1))
        assert priv == 0, f"{name}: {priv} bytes of scratch (a spill is a vector-memory operation the waits do not count)"
        assert not re.search(r"^\s*(scratch_|buffer_)", k["body"], re.M), name
    for name, k in dbl.items():
        loop = block_loop(k["body"])
        stores = re.findall(r"^\s*global_store_\w+", loop, re.M)
        # one pair-record store per block; the coarse-table instantiations also hold the two stores of the direct records
        # in their two layouts (by block / query-major) — three exclusive paths, waited for with vmcnt(1) / vmcnt(2)
        want = 5 if k["table"] else 1
        assert stores == ["\tglobal_store_dwordx4"] * want, f"{name}: stores in the block loop: {stores}"
        loads = re.findall(r"^\s*global_load_(?!lds)\w+", loop, re.M)
        assert not loads, f"{name}: plain global loads inside the block loop: {loads}"
        assert re.search(r"global_load_lds_dwordx4", loop), name
        waits = re.findall(r"s_waitcnt vmcnt\((\d+)\)", loop)
        allowed = {"0", "1", "2"} if k["table"] else {"0", "1"}
        if k["nbuf"] == 3:  # ring of three: the wait also leaves this iteration's LDS-DMA (<= 5 per wave) in flight
            allowed = {str(i) for i in range(8)}
        assert "1" in waits and set(waits) <= allowed, f"{name}: vmcnt waits {waits}"


# ---------------------------------------------------------------------------------------------------------------
# streaming rank kernel (rank_stream.hip): the default instantiations (groups of 128 queries) must not spill, and a
# step of the tile loop must multiply WITHOUT waiting for the tile it has just requested: the compiler once placed
# `s_waitcnt vmcnt(0)` in front of the first MFMA of every step (after the prefetch of the next tile), which the
# explicit tile_landed() touch moved in front of the prefetch.
# ---------------------------------------------------------------------------------------------------------------
STREAM_SRC = os.path.join(ROOT, "vector-indexer_amd", "csrc", "rank_stream.hip")


@pytest.fixture(scope="module")
def stream_asm(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not installed")
    out = str(tmp_path_factory.mktemp("isa") / "rank_stream.s")
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize"]
    subprocess.check_call([HIPCC, *flags, "--cuda-device-only", "-S", "-o", out, STREAM_SRC], stderr=subprocess.DEVNULL)
    return open(out).read()


def stream_kernels(asm_text):
    out = {}
    for m in re.finditer(r"^(_ZN2vi12_GLOBAL__N_118rank_stream_kernelILi(\d+)ELi(\d)ELb([01])ELi(\d)EEEvNS_14RankStreamArgsE):[^\n]*\n(.*?)\n\.Lfunc_end",
                         asm_text, re.S | re.M):
        name, nc, rank, qlo, nu, body = m.groups()
        meta = re.search(r"\.amdhsa_kernel " + re.escape(name) + r"\n(.*?)\.end_amdhsa_kernel", asm_text, re.S).group(1)
        out[name] = dict(body=body, meta=meta, nc=int(nc), rank=int(rank), qlo=qlo == "1", nu=int(nu))
    return out


def test_streaming_rank_kernels_do_not_spill(stream_asm):
    ks = stream_kernels(stream_asm)
    assert len(ks) == 32   # NC 1..8 x {queries hi-only, hi + lo} x {groups of 128, 256}
    for name, k in ks.items():
        if k["nu"] != 4:
            continue   # groups of 256 are an experiment knob (VI_STREAM_GQ=256); its hi + lo fallback does spill
        scratch = int(re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", k["meta"]).group(1))
        assert scratch == 0, f"{name}: {scratch} bytes of scratch per lane"
        assert not re.search(r"\bscratch_(load|store)|buffer_(load|store)", k["body"]), name


def test_streaming_rank_kernel_multiplies_while_the_next_tile_loads(stream_asm):
    ks = stream_kernels(stream_asm)
    checked = 0
    for name, k in ks.items():
        if k["nu"] != 4 or k["nc"] < 4:
            continue
        lines = k["body"].split("\n")
        tile_loads = [i for i, l in enumerate(lines) if "global_load_dwordx4" in l]
        mfmas = [i for i, l in enumerate(lines) if "v_mfma_f32_32x32x16_bf16" in l]
        # every run of tile loads that is followed by a step's MFMAs: no vmcnt wait between its last load and the step's last MFMA
        steps = 0
        for i in tile_loads:
            nxt = [m for m in mfmas if m > i]
            if not nxt or any(i < t < nxt[0] for t in tile_loads):
                continue   # not the last load of its run
            chain = [m for m in nxt if m < i + 600][: k["nc"] * (2 if k["qlo"] else 1) * 4]
            if len(chain) < k["nc"]:
                continue
            between = "\n".join(lines[i + 1:chain[-1]])
            if "s_barrier" in between:
                continue   # (the run in front of the item loop)
            # (counted waits of the conditional last-step block in between — the next item's first tile goes into registers
            # whose previous loads the compiler cannot prove complete — are harmless; a full drain is the bug)
            assert not re.search(r"s_waitcnt[^\n]*vmcnt\(0\)", between), f"{name}: a step drains vector memory after its prefetch"
            steps += 1
        assert steps >= 2, name   # the two halves of the unrolled tile loop
        checked += 1
    assert checked >= 8


# ---------------------------------------------------------------------------------------------------------------
# round 3 kernels.  (1) the k-means candidate sweep (assign_mfma.hip): no spill in any instantiation; inside the tile loop
# NO global store (a store would make the step's `s_waitcnt vmcnt(0)` wait for its round trip: measured 56 -> 45 ms at C3)
# and no plain global load (the tile arrives by LDS-DMA); the MFMA chain of a step is not interrupted by a full LDS
# drain (fragments run two chunks ahead).  (2) the big-cluster sum kernel (kmeans.hip) keeps its three register sets in
# registers (an array of HIP's float4 struct once sat in scratch memory: 5.0 -> 1.0 ms).  (3) the select kernels' wave
# top-K exchanges lanes with DPP / v_permlane*_swap: no ds_bpermute in the sorting network (wave_sort.hpp).
# ---------------------------------------------------------------------------------------------------------------
def _compile(tmp_path_factory, src):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not installed")
    out = str(tmp_path_factory.mktemp("isa") / (os.path.basename(src) + ".s"))
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize"]
    subprocess.check_call([HIPCC, *flags, "--cuda-device-only", "-S", "-o", out, src], stderr=subprocess.DEVNULL)
    return open(out).read()


def _kernel_bodies(asm_text, pattern):
    out = {}
    for m in re.finditer(r"^(" + pattern + r"):[^\n]*\n(.*?)\n\.Lfunc_end", asm_text, re.S | re.M):
        name, body = m.group(1), m.group(m.lastindex)
        mm = re.search(r"\.amdhsa_kernel " + re.escape(name) + r"\n(.*?)\.end_amdhsa_kernel", asm_text, re.S)
        out[name] = (body, mm.group(1) if mm else "")   # (device functions have no kernel descriptor)
    return out


def test_candidate_sweep_keeps_its_tile_loop_free_of_global_memory(tmp_path_factory):
    text = _compile(tmp_path_factory, os.path.join(ROOT, "vector-indexer_amd", "csrc", "assign_mfma.hip"))
    ks = _kernel_bodies(text, r"_ZN2vi12_GLOBAL__N_123mfma_assign_cand_kernelILi\d+EEEvNS0_8CandArgsE")
    assert len(ks) == 8   # NC = 1 .. 8 chunks of 16 dimensions
    for name, (body, meta) in ks.items():
        scratch = int(re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", meta).group(1))
        assert scratch == 0, f"{name}: {scratch} bytes of scratch per lane"
        lines = body.split("\n")
        barriers = [i for i, l in enumerate(lines) if "s_barrier" in l]
        stores = [i for i, l in enumerate(lines) if re.match(r"\s*global_store_", l)]
        loads = [i for i, l in enumerate(lines) if re.match(r"\s*global_load_(?!lds)", l)]
        assert any("global_load_lds_dwordx4" in l for l in lines), name
        # (whatever shape the compiler gives the tile loop: every store comes after the last barrier — the final flush of the
        # lists — and every plain load before the first one — the points' own values)
        assert stores and min(stores) > barriers[-1], f"{name}: a global store inside the sweep"
        assert loads and max(loads) < barriers[0], f"{name}: a plain global load inside the sweep"
        assert len(re.findall(r"v_mfma_f32_32x32x16_bf16", body)) >= 4 * int(re.search(r"ILi(\d+)E", name).group(1)), name
    # the D = 128 instantiation: between the first and the last MFMA of a step the LDS counter is never drained to
    # zero except at the very end (the last fragments) — i.e. the fragment reads run ahead of the chain
    body8 = [b for n, (b, _) in ks.items() if "ILi8E" in n][0]
    lines = block_loop(body8).split("\n")
    mf = [i for i, l in enumerate(lines) if "v_mfma_f32_32x32x16_bf16" in l]
    chain = lines[mf[0]:mf[31] + 1]
    drains = [i for i, l in enumerate(chain) if re.search(r"s_waitcnt[^\n]*lgkmcnt\(0\)", l)]
    assert all(i > len(chain) - 12 for i in drains), "the MFMA chain waits for every fragment read"


def test_big_cluster_sums_stay_in_registers(tmp_path_factory):
    text = _compile(tmp_path_factory, os.path.join(ROOT, "vector-indexer_amd", "csrc", "kmeans.hip"))
    ks = _kernel_bodies(text, r"_ZN2vi12_GLOBAL__N_118segment_big_kernelE\w+")
    assert len(ks) == 1
    for name, (body, meta) in ks.items():
        scratch = int(re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", meta).group(1))
        assert scratch == 0, f"{name}: {scratch} bytes of scratch per lane (the row register sets must not live in memory)"
        assert not re.search(r"\bscratch_(load|store)", body), name
        assert len(re.findall(r"global_load_dwordx4", body)) >= 8, name   # 16-byte row pieces


def test_select_top_k_sorts_without_the_lds_crossbar(asm):
    ks = _kernel_bodies(asm, r"_ZN2vi12_GLOBAL__N_113offer_bulk_fnINS_(8FastTopK|10FastTop128)EEET_S\d_fji")
    assert len(ks) == 2
    for name, (body, _) in ks.items():
        assert "ds_bpermute" not in body, f"{name}: the sorting network goes through the LDS crossbar again"
        assert "v_permlane32_swap" in body and "v_permlane16_swap" in body and "_dpp" in body, name
        assert len(re.findall(r"v_cmp_\w+_u64", body)) >= 27, name   # one 64-bit compare per compare-exchange step


def test_exact_evaluation_pieces_keep_their_address_spaces_and_counted_waits(asm):
    """The exact-evaluation pieces of the select kernels are real (noinline) functions: their pointer arguments are
    generic, and without the address-space casts every access — the query row in LDS included — is a flat_load through
    the vector-memory address pipe.  The staged row evaluation of the coarse select also counts its LDS-DMA copies by
    hand (vmcnt(4) / vmcnt(0)): correct only while those copies are the function's ONLY vector-memory operations."""
    fns = _kernel_bodies(asm, r"_ZN2vi12_GLOBAL__N_1\d+exact_batch_\w*fnINS_(8FastTopK|10FastTop128)EEET_\w+")
    assert len(fns) >= 8, sorted(fns)
    for name, (body, _) in fns.items():
        assert not re.search(r"^\s*flat_(load|store)", body, re.M), f"{name}: generic-pointer access"
        assert not re.search(r"^\s*scratch_(load|store)", body, re.M) or "u8_fn" in name, f"{name}: spills"
    staged = {n: b for n, (b, _) in fns.items() if "rows_staged" in n}
    assert len(staged) == 1
    for name, body in staged.items():
        assert len(re.findall(r"global_load_lds_dwordx4", body)) >= 12, name          # two chunks up front, one per round
        assert not re.search(r"^\s*global_load_(?!lds)", body, re.M), f"{name}: a plain global load among the counted copies"
        assert not re.search(r"^\s*(global_store|global_atomic|scratch_|buffer_)", body, re.M), name
        assert re.search(r"s_waitcnt vmcnt\(4\)", body) and re.search(r"s_waitcnt vmcnt\(0\)", body), name
        assert len(re.findall(r"v_pk_add_f32", body)) >= 8 and len(re.findall(r"v_pk_mul_f32", body)) >= 8, name
        assert not re.search(r"v_(fma|fmac|mad)_f32", body), f"{name}: a fused multiply-add in the reference's unfused chain"
