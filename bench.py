#!/usr/bin/env python3
"""bench.py — headline benchmark of the IVF search hot path on MI355X.

Metric (BASELINE.json): QPS at recall@10 >= 0.95 on SIFT1M — config C2: N=1e6, D=128, IVF k=4096, nprobe=32, search on
1 x MI355X.  SIFT1M is not available offline, so the default workload is the SIFT-shaped synthetic set of the same size
(make_dataset); `data` says so.  With --base/--query[/--gt] (.fvecs/.ivecs/.npy) the real files are used instead.

A "step" = one pass of the search hot path (coarse quantizer -> list ranking -> exact top-k) over one batch of NQ
queries that are already resident in HBM.  value = queries/s over the timed steps at BASELINE's nprobe (32).

    python bench.py [--gpus N] [--steps K] [--warmup W]

N > 1 (launched by torch.distributed.run, one rank per GPU): the same index is partitioned over the ranks (stripes:
block b of every list on rank b % N; or --placement shards: whole shard files, greedy by bytes), the coarse step is
split over the ranks by query (one all-gather of the probe lists), every rank ranks the whole batch against what it
owns, the per-rank top-k are exchanged with ONE all-gather over RCCL and merged with the reference's stable candidate
order.  Total work is fixed => "scaling": "strong".

Prints ONE JSON line (rank 0).  Besides the contract's fields it carries: roofline (dominant kernel, measured live with
HIP events on the library's stream), cpu_baseline (the oracle on the host cores, bounded sample), and `extras`: the
smallest nprobe reaching recall 0.95, the host-pointer entry, small-batch latencies, real-valued data (bf16 x 3 ranking),
config C1 on the CPU path and on the GPU, the k-means lines (C3 exact assign, update pass, reference-compat train).
"""
import argparse
import json
import os
import shutil
import sys
import tempfile
import time


# libgomp reads these once, when torch first loads it: cap the CPU baseline's OpenMP team to the CPUs
# this job may really use (the GPU box shows 256 cores to nproc but grants a 16-CPU share)
def _usable_cpus():
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, 16))


os.environ.setdefault("OMP_NUM_THREADS", str(_usable_cpus()))
os.environ.setdefault("OMP_WAIT_POLICY", "passive")

import numpy as np  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "vector-indexer_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
MFMA_F32_PEAK_TF = 157.3    # dense f32 matrix peak: 256 CUs x 256 flop/clk x 2.4 GHz
MFMA_BF16_PEAK_TF = 2516.6  # dense bf16 matrix peak: 256 CUs x 4096 flop/clk x 2.4 GHz
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "r03_scan_traffic.json")


def make_dataset(n, d, nq, seed, device):
    """SIFT-shaped synthetic data: non-negative integer-valued f32 in [0, 218] with cluster structure
    (a mixture of Gaussians, as local-descriptor sets have), generated on the GPU with a fixed seed.
    Queries are drawn from the same mixture (held-out points), like sift_query vs sift_base."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    ncomp = 2048
    centers = torch.randn(ncomp, d, generator=g, device=device) * 30.0 + 60.0

    def draw(m):
        comp = torch.randint(0, ncomp, (m,), generator=g, device=device)
        x = centers[comp] + torch.randn(m, d, generator=g, device=device) * 14.0
        return torch.clamp(torch.round(x.abs()), 0, 218).to(torch.float32).contiguous()
    return draw(n), draw(nq)


def ground_truth(xb, xq, k):
    """exact top-k by brute force on the GPU (measurement only, fp32 matmul form)."""
    import torch
    nb = xb.shape[0]
    best_d = torch.full((xq.shape[0], k), float("inf"), device=xq.device)
    best_i = torch.full((xq.shape[0], k), -1, dtype=torch.int64, device=xq.device)
    qn = (xq * xq).sum(1, keepdim=True)
    for s in range(0, nb, 131072):
        blk = xb[s:s + 131072]
        dist = qn - 2.0 * (xq @ blk.T) + (blk * blk).sum(1)[None, :]
        dd, ii = torch.topk(dist, min(k, blk.shape[0]), dim=1, largest=False)
        cat_d = torch.cat([best_d, dd], 1)
        cat_i = torch.cat([best_i, ii + s], 1)
        sel = torch.topk(cat_d, k, dim=1, largest=False)
        best_d, best_i = sel.values, torch.gather(cat_i, 1, sel.indices)
    return best_i


def recalls(I, gt):
    """(1-NN-in-top-k recall of the reference's harness, bench_all_ivf.py:336-350;
        intersection recall of the Rust tests, tests/test_utils/mod.rs:214-221)"""
    k = I.shape[1]
    r1 = (I == gt[:, :1]).any(dim=1).float().mean().item()
    inter = (I[:, :, None] == gt[:, None, :k]).any(dim=2).float().sum(dim=1).mean().item() / k
    return r1, inter


def time_steps(fn, steps, warmup, barrier, world, device):
    """W untimed + exactly K timed steps, bracketed by barrier + synchronize; MAX over the ranks"""
    import torch
    import torch.distributed as dist
    for _ in range(warmup):
        fn()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    barrier()
    el = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([el], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    return el


def roofline_of(st, scan_ms, d, world_note=""):
    """roofline of the dominant kernel (list ranking) from the library's own HIP-event time of that kernel.
    bound = "mfma": a batch turns the list scan into a dense contraction (2*D flop per (query, scanned vector) pair)
    whose floor on this chip is the matrix pipe (0.06 ms at C2), above the HBM floor of streaming the index once
    (0.04 ms).  Both fractions are printed; `traffic` = HBM-side bytes per launch from the committed PMC passes."""
    scan_s = scan_ms / 1000.0
    pairs = float(st["scanned_vectors"])
    if st["filter_tile_blocks"] > 0:
        mode = int(st["rank_mode"])  # 1 f32 MFMA, 2 bf16x3, 3 bf16 on hi planes only
        flops = 2.0 * d * pairs
        tf = flops / scan_s / 1e12 if scan_s > 0 else 0.0
        peak = MFMA_F32_PEAK_TF if mode == 1 else MFMA_BF16_PEAK_TF
        name = {1: "filter_kernel<NG,1,false,0,GQ> (f32 MFMA ranking)",
                2: "filter_kernel<NG,1,false,1,GQ> (bf16 MFMA ranking, operands split hi+lo: 3 MFMAs per 16 dims)",
                3: "rank_stream_kernel<NC,2,QLO,NU> (bf16 MFMA ranking, hi planes only: stored values are bf16-exact; queries in LDS, "
                   "vectors through registers, persistent workgroups)"}[mode]
        return {"kernel": name + " of (query group x list segment) tiles", "bound": "mfma", "achieved": round(tf, 1),
                "peak": peak, "unit": "TFLOP/s", "frac": round(tf / peak, 4), "traffic": None,
                "flops_per_launch": flops, "avg_launch_ms": round(scan_ms, 4), "tiles_per_launch": int(st["filter_tile_blocks"]),
                "queries_per_work_item": int(st["group_queries"]) or 128,
                "peak_note": ("dense f32 MFMA peak" if mode == 1 else "dense bf16 MFMA peak (no sparsity)") +
                             "; useful flops = 2*D per (query, scanned vector) pair" +
                             (", the 3 split products of bf16x3 are overhead, not counted" if mode == 2 else "") + world_note}
    algo = pairs * (4 * d + 8)  # 4*D per vector + 8 B id, no reuse
    gbs = algo / scan_s / 1e9 if scan_s > 0 else 0.0
    return {"kernel": "scan_kernel<LISTS> (exact-order VALU list scan + wave top-k)", "bound": "hbm", "achieved": round(gbs, 1),
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": None,
            "avg_launch_ms": round(scan_ms, 4), "algorithmic_bytes_per_launch": int(algo)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--n", type=int, default=1_000_000)
    ap.add_argument("--d", type=int, default=128)
    ap.add_argument("--nlist", type=int, default=4096)
    ap.add_argument("--nq", type=int, default=10_000)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--nprobe", type=int, default=32, help="headline operating point (BASELINE configs[1]: 32)")
    ap.add_argument("--base", default=None, help="database vectors (.fvecs / .npy), e.g. sift_base.fvecs")
    ap.add_argument("--query", default=None, help="query vectors (.fvecs / .npy)")
    ap.add_argument("--gt", default=None, help="ground truth (.ivecs / .npy); computed exactly when absent")
    ap.add_argument("--placement", choices=["stripes", "shards"], default="stripes")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kmeans", action="store_true")
    ap.add_argument("--no-real-valued", action="store_true", help="skip the extra index of real-valued (noisy) data")
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--assign-n", type=int, default=10_000_000)
    ap.add_argument("--work-dir", default=None)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product has no CPU fallback")
    # rehearsal on a one-GPU box: VI_BENCH_ONE_DEVICE=1 puts every rank on GPU 0 and exchanges over gloo (RCCL
    # refuses two ranks on one device); the driver's real N-GPU runs use one GPU per rank and RCCL
    rehearsal = os.environ.get("VI_BENCH_ONE_DEVICE") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)

    import ctypes as C
    import vector_indexer_py as vip
    from vector_indexer_py import _native, harness

    lib = _native.lib()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- data + index -------------------------------------------------------------------------
    gt = None
    if args.base:
        xb_h = harness.load_vectors(args.base)
        xq_h = harness.load_vectors(args.query)[:args.nq]
        xb, xq = torch.from_numpy(xb_h).to(device), torch.from_numpy(xq_h).to(device)
        args.n, args.d = xb.shape
        data = f"{os.path.basename(args.base)} / {os.path.basename(args.query)}"
        if args.gt:
            gt = torch.from_numpy(harness.load_groundtruth(args.gt)[:xq.shape[0], :args.k]).to(device)
    else:
        xb, xq = make_dataset(args.n, args.d, args.nq, 42, device)
        data = "synthetic (SIFT1M-shaped mixture, seed 42; SIFT1M itself is not available offline)"
    nq, k, d = xq.shape[0], args.k, args.d
    work = args.work_dir or os.path.join(tempfile.gettempdir(), f"vi_bench_{os.getuid()}_{args.n}_{d}_{args.nlist}")
    build = None
    if rank == 0:
        shutil.rmtree(work, ignore_errors=True)
        t0 = time.time()
        built = vip.build(xb.cpu().numpy(), work, nlist=args.nlist, now_secs=1_700_000_000, device=local_rank)
        bs = built.build_stats()
        build = {"seconds": round(time.time() - t0, 2), "phases_ms": {p[3:]: round(bs[p], 1) for p in
                 ("ms_upload", "ms_kmeans", "ms_group", "ms_super", "ms_export", "ms_index")},
                 "shard_bytes": int(bs["shard_bytes"]), "lists": int(bs["lists"]), "shards": int(bs["shards"])}
        del built
    barrier()
    placement = 1 if (args.placement == "shards" and world > 1) else 0
    index = vip.load(os.path.join(work, "index"), os.path.join(work, "shards"), d, device=local_rank,
                     rank=rank if world > 1 else 0, world_size=world if world > 1 else 0, placement=placement)
    index.enable_timing(True)
    D = torch.empty((nq, k), dtype=torch.float32, device=device)
    I = torch.empty((nq, k), dtype=torch.int64, device=device)
    T = torch.empty((nq, k), dtype=torch.int64, device=device)
    if world > 1:
        # per-rank results packed [D | I | tie] so that ONE all-gather exchanges them; every buffer of a step is
        # allocated once here (per probe count), nothing inside the step allocates or copies through the host
        S = int(lib.vi_packed_result_bytes(nq, k))
        off_i = (nq * k * 4 + 7) // 8 * 8
        off_t = off_i + nq * k * 8
        mine = torch.empty(S, dtype=torch.uint8, device=device)
        gathered = torch.empty(world * S, dtype=torch.uint8, device=device)
        Dm = torch.empty((nq, k), dtype=torch.float32, device=device)
        Im = torch.empty((nq, k), dtype=torch.int64, device=device)
        per = (nq + world - 1) // world
        q0, q1 = min(nq, rank * per), min(nq, (rank + 1) * per)
        probe_bufs = {}

        def bufs_for(p_eff):
            if p_eff not in probe_bufs:
                po = torch.full((2, per, p_eff), -1, dtype=torch.int32, device=device)          # [probes | order] of my slice
                pog = torch.empty((world, 2, per, p_eff), dtype=torch.int32, device=device)
                pa = torch.empty((world * per, p_eff), dtype=torch.int32, device=device)
                oa = torch.empty((world * per, p_eff), dtype=torch.int32, device=device)
                probe_bufs[p_eff] = (po, pog, pa, oa)
            return probe_bufs[p_eff]

    xq_p, D_p, I_p, T_p = xq.data_ptr(), D.data_ptr(), I.data_ptr(), T.data_ptr()

    def step(n_probe):
        if world == 1:
            index.search_device(xq_p, nq, k, n_probe, D_p, I_p, T_p)
            return I
        # The library works on its own blocking stream: it starts after what torch queued on the default stream
        # (incl. the wait for a collective) and returns when its own work is done — no explicit synchronisation here.
        p_eff = min(n_probe, index.num_centroids)
        po, pog, pa, oa = bufs_for(p_eff)
        if q1 > q0:
            index.probe_device(xq[q0:].data_ptr(), q1 - q0, n_probe, po[0].data_ptr(), po[1].data_ptr())
        dist.all_gather_into_tensor(pog.view(-1), po.view(-1))  # (flat: the form both RCCL and gloo accept)
        pa.view(world, per, p_eff).copy_(pog[:, 0])
        oa.view(world, per, p_eff).copy_(pog[:, 1])
        base = mine.data_ptr()
        index.search_probed_device(xq.data_ptr(), nq, k, p_eff, pa.data_ptr(), oa.data_ptr(), base, base + off_i,
                                   base + off_t)
        dist.all_gather_into_tensor(gathered, mine)
        torch.cuda.current_stream().synchronize()  # the merge below runs on the null stream of the library's device
        _native.check(lib.vi_merge_partials_packed_device(local_rank, nq, k, world, gathered.data_ptr(), Dm.data_ptr(),
                                                          Im.data_ptr()))
        return Im

    # ---- recall at every operating point of the sweep (scripts/run_faiss_bench.sh:55) -----------------
    if gt is None:
        gt = ground_truth(xb, xq, k)
    sweep = {}
    min_ok = None
    for p in sorted({1, 2, 4, 8, 16, 32, 64, args.nprobe}):
        r1, ri = recalls(step(p), gt)
        sweep[p] = {"recall_1nn_at_k": round(r1, 4), "recall_at_k": round(ri, 4)}
        if min_ok is None and ri >= 0.95:
            min_ok = p
    head = args.nprobe

    # ---- timed region: the headline operating point ---------------------------------------------------
    # Inside the timed region the library records HIP events around the dominant kernel only (the list rank: what the
    # roofline is priced on); an event at every phase boundary is five barrier packets per step, 0.01 ms of 0.5.  The
    # phase breakdown comes from a few more steps right after, outside the timed region.
    scan_ms, coarse_ms, group_ms, sel_ms, tot_ms = [], [], [], [], []
    index.enable_timing(2)

    def timed_step():
        step(head)
        scan_ms.append(index.last_stat("ms_scan"))
    elapsed = time_steps(timed_step, args.steps, args.warmup, barrier, world, device)
    scan_ms = scan_ms[args.warmup:]
    st = index.last_stats()
    ms_per_step = elapsed * 1000.0 / args.steps
    qps = nq * args.steps / elapsed
    index.enable_timing(True)
    for i in range(2 + max(3, args.steps // 2)):
        step(head)
        sp = index.last_stats()
        if i >= 2:
            coarse_ms.append(sp["ms_coarse"]); group_ms.append(sp["ms_group"]); sel_ms.append(sp["ms_merge"]); tot_ms.append(sp["ms_total"])
    roofline = roofline_of(st, float(np.mean(scan_ms)), d)
    roofline["pipeline_ms"] = {"coarse": round(float(np.mean(coarse_ms)), 4), "grouping": round(float(np.mean(group_ms)), 4),
                               "list_rank": round(float(np.mean(scan_ms)), 4), "select": round(float(np.mean(sel_ms)), 4),
                               "total": round(float(np.mean(tot_ms)), 4),
                               "note": "list_rank: HIP events inside the timed region; the other phases from steps run right after it "
                                       "with an event at every phase boundary"}
    # algorithmic bytes of one launch: the resident image streamed once + the queries + the records written
    image_bytes = index.num_vectors * d * (2 if int(st["rank_mode"]) == 3 else (4 if int(st["rank_mode"]) == 2 else 4))
    algo_bytes = image_bytes + nq * d * 4 + st["filter_tile_blocks"] * 8 * 2 * (int(st["group_queries"]) or 128)
    scan_s = float(np.mean(scan_ms)) / 1000.0
    roofline["algorithmic_bytes_per_launch"] = int(algo_bytes)
    roofline["frac_hbm_algorithmic"] = round(algo_bytes / scan_s / 1e9 / HBM_PEAK_GBS, 4) if scan_s > 0 else None
    # HBM-side traffic of that kernel from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE of this same
    # command; gfx950 correction as MI355X_MICROARCH.md prescribes), when they were taken on this workload and kernel
    try:
        tr = json.load(open(TRAFFIC_FILE))
        if tr["workload"] == [args.n, d, args.nlist, head, nq, k] and world == 1 and tr.get("rank_mode") == int(st["rank_mode"]):
            roofline["traffic"] = tr["hbm_bytes_per_launch"]
            roofline["traffic_source"] = tr["source"]
            roofline["frac_hbm_traffic"] = round(tr["hbm_bytes_per_launch"] / scan_s / 1e9 / HBM_PEAK_GBS, 4)
            roofline["traffic_over_algorithmic"] = round(tr["hbm_bytes_per_launch"] / algo_bytes, 2)
    except Exception:
        pass

    extras = {}
    # ---- multi-GPU: how the scan work splits under the two placements (share of the busiest rank) ----
    if world > 1:
        sv = torch.tensor([float(st["scanned_vectors"])], dtype=torch.float64, device=device)
        allsv = [torch.zeros_like(sv) for _ in range(world)]
        dist.all_gather(allsv, sv)
        tot = sum(float(x.item()) for x in allsv)
        extras["scan_share_of_busiest_rank"] = {args.placement: round(max(float(x.item()) for x in allsv) / max(tot, 1.0), 4),
                                                "ideal": round(1.0 / world, 4)}
        # ... and under the OTHER partition rule, from one untimed step on a second handle (same files, same probe lists)
        other = "shards" if args.placement == "stripes" else "stripes"
        alt = vip.load(os.path.join(work, "index"), os.path.join(work, "shards"), d, device=local_rank, rank=rank, world_size=world,
                       placement=1 if other == "shards" else 0)
        alt.enable_timing(True)
        p_eff = min(head, index.num_centroids)
        po, pog, pa, oa = bufs_for(p_eff)
        alt.search_probed_device(xq.data_ptr(), nq, k, p_eff, pa.data_ptr(), oa.data_ptr(), mine.data_ptr(), mine.data_ptr() + off_i,
                                 mine.data_ptr() + off_t)
        sv2 = torch.tensor([float(alt.last_stats()["scanned_vectors"])], dtype=torch.float64, device=device)
        allsv2 = [torch.zeros_like(sv2) for _ in range(world)]
        dist.all_gather(allsv2, sv2)
        tot2 = sum(float(x.item()) for x in allsv2)
        extras["scan_share_of_busiest_rank"][other] = round(max(float(x.item()) for x in allsv2) / max(tot2, 1.0), 4)
        del alt

    if rank == 0 and world == 1 and not args.no_extras:
        # (a) the smallest nprobe of the sweep reaching recall@10 >= 0.95, timed the same way
        if min_ok is not None and min_ok != head:
            e = time_steps(lambda: step(min_ok), args.steps, args.warmup, barrier, world, device)
            extras["min_nprobe_at_recall_0.95"] = {"nprobe": min_ok, "queries_per_s": round(nq * args.steps / e, 1),
                                                   "ms_per_step": round(e * 1000.0 / args.steps, 4), **sweep[min_ok]}
        # (b) host-pointer entry (vi_indexer_search: queries and results cross PCIe inside the call)
        xq_h, Dh, Ih = xq.cpu().numpy(), np.empty((nq, k), np.float32), np.empty((nq, k), np.int64)
        kout = C.c_uint64(0)

        def host_step():
            _native.check(lib.vi_indexer_search(index._h, _native.ptr(xq_h), nq, d, k, head, _native.ptr(Dh), _native.ptr(Ih),
                                                None, None, C.byref(kout)))
        e = time_steps(host_step, max(3, args.steps // 2), 2, barrier, world, device)
        extras["host_pointer_entry"] = {"queries_per_s": round(nq * max(3, args.steps // 2) / e, 1),
                                        "ms_per_step": round(e * 1000.0 / max(3, args.steps // 2), 4),
                                        "note": "vi_indexer_search: 5 MB of queries H2D + results D2H per call, PCIe-inclusive"}
        # (c) small batches: latency per call at the headline nprobe
        lat = {}
        for b in (1, 64, 1024):
            if b <= nq:
                def small():
                    index.search_device(xq.data_ptr(), b, k, head, D.data_ptr(), I.data_ptr(), 0)
                e = time_steps(small, 50, 5, barrier, world, device)
                lat[str(b)] = {"ms_per_call": round(e * 1000.0 / 50, 4), "queries_per_s": round(b * 50 / e, 1)}
        extras["batch_latency"] = lat
        # (c2) several host threads issuing batches on the one handle (the library holds four search contexts, each with its
        # own stream: the reference's search is &self and runs concurrently, tests/ivf_index_tests.rs:768-807): batches
        # overlap on the GPU — a serving process's throughput, not the headline (whose steps run one after the other)
        import threading
        conc = {}
        index.search_device(xq.data_ptr(), nq, k, head, D.data_ptr(), I.data_ptr(), 0)  # single-thread reference for the comparison
        torch.cuda.synchronize(device)
        for nthreads in (2, 4):
            outs = [(torch.empty_like(D), torch.empty_like(I)) for _ in range(nthreads)]
            per = max(4, args.steps // 2)

            def worker(t, reps):
                for _ in range(reps):
                    index.search_device(xq.data_ptr(), nq, k, head, outs[t][0].data_ptr(), outs[t][1].data_ptr(), 0)
            for t in range(nthreads):
                worker(t, 1)
            torch.cuda.synchronize(device)
            t0 = time.perf_counter()
            th = [threading.Thread(target=worker, args=(t, per)) for t in range(nthreads)]
            for x in th:
                x.start()
            for x in th:
                x.join()
            torch.cuda.synchronize(device)
            e = time.perf_counter() - t0
            conc[str(nthreads)] = {"queries_per_s": round(nthreads * per * nq / e, 1), "ms_per_batch": round(e * 1000.0 / (nthreads * per), 4),
                                   "identical_to_single_thread": bool(all(torch.equal(o[1], I) for o in outs))}
        extras["concurrent_batches_one_handle"] = conc
        # (d) real-valued data takes the bf16 x 3 ranking (lo planes streamed): the same batch with that arithmetic
        os.environ["VI_FILTER_HI_ONLY"] = "0"
        e = time_steps(lambda: step(head), max(3, args.steps // 2), 2, barrier, world, device)
        st3 = index.last_stats()
        del os.environ["VI_FILTER_HI_ONLY"]
        extras["bf16x3_ranking"] = {"queries_per_s": round(nq * max(3, args.steps // 2) / e, 1),
                                    "ms_per_step": round(e * 1000.0 / max(3, args.steps // 2), 4),
                                    "list_rank_ms": round(st3["ms_scan"], 4), "rank_mode": int(st3["rank_mode"]),
                                    "note": "what real-valued (not bf16-exact) stored vectors run; ids and distances are the same bits"}

        # (e) genuinely real-valued lists far from the origin: the same data with U(0, 1/2) noise on every value (nothing
        # is bf16-exact any more).  The library takes the ranking images about the mean of the stored vectors and, spread
        # permitting, ranks from the hi planes alone (rank_mode 5 / 6); the forced bf16 x 3 run must return the same bits.
        if not args.no_real_valued:
            gn = torch.Generator(device=device)
            gn.manual_seed(7)
            xbr = xb + torch.rand(xb.shape, generator=gn, device=device) * 0.5
            xqr = (xq + torch.rand(xq.shape, generator=gn, device=device) * 0.5).contiguous()
            work_r = work + "_real"
            shutil.rmtree(work_r, ignore_errors=True)
            idx_r = vip.build(xbr.cpu().numpy(), work_r, nlist=args.nlist, now_secs=1_700_000_000, device=local_rank)
            del xbr
            idx_r.enable_timing(True)
            Dr, Ir = torch.empty_like(D), torch.empty_like(I)
            reps = max(3, args.steps // 2)

            def step_r():
                idx_r.search_device(xqr.data_ptr(), nq, k, head, Dr.data_ptr(), Ir.data_ptr(), 0)
            e = time_steps(step_r, reps, 2, barrier, world, device)
            str_ = idx_r.last_stats()
            Dk, Ik = Dr.clone(), Ir.clone()
            os.environ["VI_RANK_APPROX"] = "0"
            os.environ["VI_FILTER_HI_ONLY"] = "0"
            e3 = time_steps(step_r, reps, 2, barrier, world, device)
            st3r = idx_r.last_stats()
            del os.environ["VI_RANK_APPROX"], os.environ["VI_FILTER_HI_ONLY"]
            extras["real_valued_lists"] = {
                "data": "the bench data + U(0, 1/2) noise on every stored and query value",
                "queries_per_s": round(nq * reps / e, 1), "ms_per_step": round(e * 1000.0 / reps, 4),
                "pipeline_ms": {"coarse": round(str_["ms_coarse"], 4), "grouping": round(str_["ms_group"], 4),
                                "list_rank": round(str_["ms_scan"], 4), "select": round(str_["ms_merge"], 4)},
                "rank_mode": int(str_["rank_mode"]),
                "forced_bf16x3": {"ms_per_step": round(e3 * 1000.0 / reps, 4), "list_rank_ms": round(st3r["ms_scan"], 4),
                                  "rank_mode": int(st3r["rank_mode"])},
                "identical_to_forced_bf16x3": bool(torch.equal(Dk.view(torch.int32), Dr.view(torch.int32)) and torch.equal(Ik, Ir))}
            del idx_r
            shutil.rmtree(work_r, ignore_errors=True)

    # ---- second BASELINE metric: k-means on the matrix cores (config C3) ---------------------------
    kmeans = None
    if rank == 0 and world == 1 and not args.no_kmeans:
        del xb
        torch.cuda.empty_cache()
        n3, d3, k3 = args.assign_n, 128, 16384
        g = torch.Generator(device=device)
        g.manual_seed(42)
        X3 = torch.randn(n3, d3, generator=g, device=device)
        C3 = X3[torch.randperm(n3, generator=g, device=device)[:k3]].contiguous()
        lab = torch.empty(n3, dtype=torch.int32, device=device)
        torch.cuda.synchronize()
        ast = _native.AssignStats()
        best = None
        for _ in range(2):
            _native.check(lib.vi_assign_device(local_rank, X3.data_ptr(), n3, d3, C3.data_ptr(), k3, 42, 1, lab.data_ptr(),
                                               C.byref(ast)))
            if best is None or ast.ms_total < best[0]:
                best = (ast.ms_total, ast.ms_filter, int(ast.ambiguous_rows), int(ast.tier1_rows))
        flops = 2.0 * n3 * k3 * d3
        tf = flops / (best[1] * 1e-3) / 1e12
        bf16 = os.environ.get("VI_ASSIGN_BF16", "1") != "0"
        cand = bf16 and os.environ.get("VI_ASSIGN_CAND", "1") != "0"
        if cand:
            kname = ("mfma_assign_cand_kernel<8> (v_mfma_f32_32x32x16_bf16 on the hi planes: ONE MFMA per product; every centroid within the "
                     "error margin of a point's running minimum is listed and decided by exact lane-order distances)")
        elif bf16:
            kname = "mfma_assign_bf16_kernel<16> (v_mfma_f32_32x32x16_bf16, operands split hi+lo: 3 products per multiply)"
        else:
            kname = "mfma_assign_kernel<16,1> (v_mfma_f32_32x32x2_f32)"
        kmeans = {"workload": f"exact nearest-centroid assign N={n3} D={d3} k={k3} (BASELINE config C3), one full pass",
                  "ms_total": round(best[0], 2), "ms_mfma_filter": round(best[1], 2),
                  "rows_left_by_first_tier": best[3], "rows_re_evaluated_exactly": best[2],
                  "roofline": {"kernel": kname,
                               "bound": "mfma", "achieved": round(tf, 1), "peak": MFMA_BF16_PEAK_TF if bf16 else MFMA_F32_PEAK_TF,
                               "unit": "TFLOP/s", "frac": round(tf / (MFMA_BF16_PEAK_TF if bf16 else MFMA_F32_PEAK_TF), 4),
                               "flops_per_launch": flops,
                               "frac_of_bf16_peak_over_3": round(tf / (MFMA_BF16_PEAK_TF / 3.0), 4) if (bf16 and not cand) else None,
                               "frac_of_f32_mfma_peak": round(tf / MFMA_F32_PEAK_TF, 4),
                               "peak_note": "achieved = useful 2*N*k*D flop / time of the first-tier kernel (HIP events on the library's "
                                            "stream); frac against the dense bf16 peak.  The sweep issues 1 + 16/256 MFMAs per product "
                                            "(its first 16 centroid tiles are swept twice); ms_total adds the exact evaluation of the "
                                            "listed candidates and the rows the lists could not hold"},
                  "hbm_GBps": round((4.0 * n3 * d3 + 4.0 * n3) / (best[1] * 1e-3) / 1e9, 1),
                  "note": "labels == assign_points_brute_force on every row the first tier leaves undecided and 20 000 sampled rows "
                          "at this k and D (tests/test_baseline_configs_gpu.py::test_c3_exact_assign_k16384, N=1e6); the whole training "
                          "run at N=1e7 against the oracle in test_c3_mini_batch_train_n1e7_k16384"}
        # CPU baseline of this metric: the oracle's assign in the reference's own mode (2-level hierarchy above 100 centroids,
        # src/kmeans.rs:445-581 — what BASELINE.md promises) and its brute force (what the GPU number computes), on a bounded
        # sample of the same points against the same centroids
        if not args.no_cpu_baseline:
            import oracle_lib as O
            threads = min(O.lib().orc_max_threads(), O.usable_cpus())
            Ch = C3.cpu().numpy()
            cb = {}
            for mode, m0 in (("hier", 200_000), ("brute", 4_000)):
                Xs = X3[:m0].cpu().numpy()
                t0 = time.perf_counter()
                lo = O.assign(Xs, Ch, seed=42, mode=mode)
                dt = time.perf_counter() - t0
                cb[mode] = {"points_per_s": round(m0 / dt, 1), "sample_points": m0, "seconds": round(dt, 2)}
                if mode == "brute":
                    cb[mode]["identical_to_gpu_labels"] = bool((lab[:m0].cpu().numpy().astype(np.uint64) == lo).all())
            kmeans["cpu_baseline"] = {"value": cb["hier"]["points_per_s"], "unit": "points/s", "cores": int(threads), "kind": "port",
                                      "sample": f"oracle assign_points_hierarchical on the first {cb['hier']['sample_points']} of the {n3} points, "
                                                f"k={k3} D={d3} (the reference's mode above 100 centroids: approximate); exact brute force on "
                                                f"{cb['brute']['sample_points']} points: {cb['brute']['points_per_s']} points/s, labels identical to the "
                                                f"GPU's: {cb['brute']['identical_to_gpu_labels']}",
                                      "gpu_points_per_s": round(n3 / (best[0] * 1e-3), 1)}
        # update pass (update_centroids_parallel): per-cluster sums + counts of all points, device-resident
        sums = torch.empty((k3, d3), dtype=torch.float32, device=device)
        cnts = torch.empty(k3, dtype=torch.int32, device=device)
        torch.cuda.synchronize()
        tu = []
        for _ in range(2):
            t0 = time.perf_counter()
            _native.check(lib.vi_kmeans_partial_sums_device(local_rank, X3.data_ptr(), n3, d3, lab.data_ptr(), k3,
                                                            sums.data_ptr(), cnts.data_ptr()))
            tu.append(time.perf_counter() - t0)
        ub = 4.0 * n3 * d3 + 8.0 * n3 + 4.0 * k3 * d3
        kmeans["update_pass"] = {"ms": round(min(tu) * 1e3, 2), "algorithmic_GBps": round(ub / min(tu) / 1e9, 1),
                                 "frac_of_hbm_peak": round(ub / min(tu) / 1e9 / HBM_PEAK_GBS, 4),
                                 "note": "vi_kmeans_partial_sums_device: grouping of the ids by cluster (stable device radix sort) + "
                                         "sums in ascending id order (the reference's order), whole call"}
        del sums, cnts, lab
        # reference-compat mini-batch train (B=256, 20 iterations + final assign), device-resident points
        Cout = torch.empty((k3, d3), dtype=torch.float32, device=device)
        lab = torch.empty(n3, dtype=torch.int32, device=device)
        it = C.c_uint64(0)
        tr_ = {}
        for name, mode in (("reference_assign_mode", 0), ("exact_assign_mode", 1)):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            _native.check(lib.vi_kmeans_mini_batch_device(local_rank, X3.data_ptr(), n3, d3, k3, 20, -1.0, 42, mode,
                                                          Cout.data_ptr(), lab.data_ptr(), C.byref(it)))
            tr_[name] = {"seconds": round(time.perf_counter() - t0, 2), "iterations": int(it.value)}
        kmeans["mini_batch_train"] = {**tr_, "note": "vi_kmeans_mini_batch_device, points resident in HBM; dominated by what the "
                                      "reference's semantics keep sequential: 16 384 k-means++ draws (each a left-to-right f32 prefix sum "
                                      "over 50 000 weights on the host, 54 us) and the draw scan of a full Fisher-Yates shuffle of 0..N "
                                      "per iteration (kmeans.rs:722-726: 27 ms; keystream from the GPU, only the batch's 256 entries traced)"}
        del X3, C3, lab, Cout
        torch.cuda.empty_cache()

    # ---- CPU baseline: the oracle (C restatement of the reference's CPU path) on the host cores -------
    cpu = None
    c1 = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import oracle_lib as O
        orc = O.OracleIndex.load(os.path.join(work, "index"), os.path.join(work, "shards"))
        threads = min(O.lib().orc_max_threads(), O.usable_cpus())
        xq_h = xq.cpu().numpy()
        done, t0c, chunk = 0, time.perf_counter(), 500
        checked = 0
        while time.perf_counter() - t0c < args.cpu_seconds and done < nq:
            m = min(chunk, nq - done)
            rc, Do, Io = orc.search_batch(xq_h[done:done + m], k, head, threads)
            if done == 0:  # parity spot check on the same queries: ids and distance bits
                Ig = step(head)[:m].cpu().numpy()
                Dg = D[:m].cpu().numpy()
                assert (Io == Ig).all() and (Do.view(np.uint32) == Dg.view(np.uint32)).all(), "GPU differs from the CPU oracle"
                checked = m
            done += m
        cpu_s = time.perf_counter() - t0c
        cpu = {"value": round(done / cpu_s, 1), "unit": "queries/s", "cores": int(threads), "kind": "port",
               "sample": f"{done} of the {nq} bench queries, same index files, nprobe={head}, k={k}, lists preloaded in RAM "
                         f"(the reference additionally re-reads shard files per query); first {checked} checked bit for bit "
                         f"against the GPU; host has {os.cpu_count()} logical cores"}
        del orc
        # config C1 (the reference's own CPU-runnable case): N=50 000 D=64 nlist=100 nprobe=8 k=10, 1 000 queries
        if not args.no_extras:
            rng = np.random.default_rng(42)
            xb1 = rng.standard_normal((50_000, 64)).astype(np.float32)
            xq1 = rng.standard_normal((1_000, 64)).astype(np.float32)
            w1 = work + "_c1"
            shutil.rmtree(w1, ignore_errors=True)
            os.makedirs(w1 + "/index"), os.makedirs(w1 + "/shards")
            t0 = time.perf_counter()
            o1 = O.OracleIndex.build(xb1, w1 + "/index", w1 + "/shards", nlist=100, now=1_700_000_000)
            tb = time.perf_counter() - t0
            t0 = time.perf_counter()
            reps = 0
            while time.perf_counter() - t0 < 3.0:
                rc, Do, Io = o1.search_batch(xq1, 10, 8, threads)
                reps += 1
            ts = (time.perf_counter() - t0) / reps
            g1 = vip.load(w1 + "/index", w1 + "/shards", 64, device=local_rank)
            Dg, Ig = g1.search_sync(xq1, 10, 8)
            t0 = time.perf_counter()
            for _ in range(20):
                g1.search_sync(xq1, 10, 8)
            tg = (time.perf_counter() - t0) / 20
            c1 = {"workload": "C1: N=50000 D=64 nlist=100 nprobe=8 k=10, 1000 queries (default_rng(42).standard_normal)",
                  "cpu_path": {"queries_per_s": round(1000 / ts, 1), "build_s": round(tb, 2), "cores": int(threads), "kind": "port"},
                  "gpu_path_host_pointers": {"queries_per_s": round(1000 / tg, 1)},
                  "identical_results": bool((Ig == Io).all() and (Dg.view(np.uint32) == Do.view(np.uint32)).all())}
            shutil.rmtree(w1, ignore_errors=True)
    if c1:
        extras["config_C1"] = c1

    if rank == 0:
        out = {"metric": "QPS at recall@10>=0.95 (IVF search, SIFT1M-shaped)", "value": round(qps, 1),
               "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
               "scaling": "strong" if world > 1 else "weak", "vs_baseline": None, "dtype": "f32", "data": data,
               "config": {"workload": f"IVF search N={args.n} D={d} nlist={args.nlist} nprobe={head} k={k} nq/step={nq} "
                                      "(BASELINE configs[1])", "nprobe": head, "recall": sweep[head], "nprobe_sweep": sweep,
                          "index_centroids": index.num_centroids, "build": build,
                          "parallelism": (f"{args.placement} over {world} GPUs, coarse table replicated, coarse step split by query, "
                                          "RCCL all-gather of probe lists and of per-rank top-k") if world > 1 else "1 GPU"},
               "roofline": roofline, "cpu_baseline": cpu, "kmeans_assign": kmeans, "extras": extras}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
