#!/usr/bin/env python3
"""bench.py — headline benchmark of the IVF search hot path on MI355X.

Metric (BASELINE.json): QPS at recall@10 >= 0.95 on SIFT1M (N=1e6, D=128, IVF k=4096, nprobe sweep,
headline nprobe=32), search on 1 x MI355X.  SIFT1M is not available offline, so the workload is the
documented SIFT-shaped synthetic set of the same size (see make_dataset); `data` says so.

A "step" = one pass of the search hot path (coarse quantizer -> list scan -> top-k) over one batch
of NQ queries that are already resident in HBM.  value = queries/s over the timed steps.

    python bench.py [--gpus N] [--steps K] [--warmup W]

N > 1 (launched by torch.distributed.run, one rank per GPU): the same index is partitioned by
block over the ranks (block b of every list on rank b % N), every rank searches the full query batch against the
lists it owns, the per-rank top-k are exchanged with ONE all-gather over RCCL (after one all-gather of the probe
lists: the coarse step is split over the ranks by query) and merged with the
reference's stable candidate order.  Total work is fixed => "scaling": "strong".

Prints ONE JSON line (rank 0).
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

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
MFMA_F32_PEAK_TF = 157.3  # dense f32 matrix peak: 256 CUs x 256 flop/clk x 2.4 GHz
MFMA_BF16_PEAK_TF = 2516.6  # dense bf16 matrix peak: 256 CUs x 4096 flop/clk x 2.4 GHz


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
        dd, ii = torch.topk(dist, k, dim=1, largest=False)
        cat_d = torch.cat([best_d, dd], 1)
        cat_i = torch.cat([best_i, ii + s], 1)
        sel = torch.topk(cat_d, k, dim=1, largest=False)
        best_d, best_i = sel.values, torch.gather(cat_i, 1, sel.indices)
    return best_i


def recalls(I, gt):
    """(1-NN-in-top-k recall of the reference's harness, bench_all_ivf.py:336-350;
        intersection recall of the Rust tests, tests/test_utils/mod.rs:214-221)"""
    import torch
    k = I.shape[1]
    r1 = (I == gt[:, :1]).any(dim=1).float().mean().item()
    inter = (I[:, :, None] == gt[:, None, :k]).any(dim=2).float().sum(dim=1).mean().item() / k
    return r1, inter


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
    ap.add_argument("--nprobe", type=int, default=0, help="0 = smallest of the sweep with recall@10 >= 0.95")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kmeans", action="store_true")
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

    import vector_indexer_py as vip
    from vector_indexer_py import _native

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- data + index -------------------------------------------------------------------------
    xb, xq = make_dataset(args.n, args.d, args.nq, 42, device)
    work = args.work_dir or os.path.join(tempfile.gettempdir(), f"vi_bench_{os.getuid()}_{args.n}_{args.d}_{args.nlist}")
    t0 = time.time()
    if rank == 0:
        shutil.rmtree(work, ignore_errors=True)
        built = vip.build(xb.cpu().numpy(), work, nlist=args.nlist, now_secs=1_700_000_000, device=local_rank)
        del built
    build_s = time.time() - t0
    barrier()
    index = vip.load(os.path.join(work, "index"), os.path.join(work, "shards"), args.d, device=local_rank,
                     rank=rank if world > 1 else 0, world_size=world if world > 1 else 0)
    index.enable_timing(True)
    nq, k = args.nq, args.k
    D = torch.empty((nq, k), dtype=torch.float32, device=device)
    I = torch.empty((nq, k), dtype=torch.int64, device=device)
    T = torch.empty((nq, k), dtype=torch.int64, device=device)
    if world > 1:
        # per-rank results packed [D | I | tie] so that ONE all-gather exchanges them
        S = int(_native.lib().vi_packed_result_bytes(nq, k))
        off_i = (nq * k * 4 + 7) // 8 * 8
        off_t = off_i + nq * k * 8
        mine = torch.empty(S, dtype=torch.uint8, device=device)
        gathered = torch.empty(world * S, dtype=torch.uint8, device=device)
        Dm = torch.empty((nq, k), dtype=torch.float32, device=device)
        Im = torch.empty((nq, k), dtype=torch.int64, device=device)

    if world > 1:  # coarse step split over the ranks by query: each rank probes its slice, one all-gather of the probes
        per = (nq + world - 1) // world
        q0, q1 = min(nq, rank * per), min(nq, (rank + 1) * per)
        pmax = 64
        PR = torch.full((per, 2 * pmax), -1, dtype=torch.int32, device=device)       # [probes | order] of my slice
        PRg = torch.empty((world * per, 2 * pmax), dtype=torch.int32, device=device)
        probes_all = torch.empty((world * per, pmax), dtype=torch.int32, device=device)
        order_all = torch.empty((world * per, pmax), dtype=torch.int32, device=device)

    def step(n_probe):
        if world == 1:
            index.search_device(xq.data_ptr(), nq, k, n_probe, D.data_ptr(), I.data_ptr(), T.data_ptr())
            return I
        p_eff = min(n_probe, index.num_centroids)
        pr = torch.empty((per, p_eff), dtype=torch.int32, device=device)
        od = torch.empty((per, p_eff), dtype=torch.int32, device=device)
        if q1 > q0:
            index.probe_device(xq[q0:].data_ptr(), q1 - q0, n_probe, pr.data_ptr(), od.data_ptr())
        PR[:, :p_eff] = pr
        PR[:, pmax:pmax + p_eff] = od
        dist.all_gather_into_tensor(PRg, PR)
        pa = PRg[:nq, :p_eff].contiguous()
        oa = PRg[:nq, pmax:pmax + p_eff].contiguous()
        torch.cuda.synchronize()
        base = mine.data_ptr()
        index.search_probed_device(xq.data_ptr(), nq, k, p_eff, pa.data_ptr(), oa.data_ptr(), base, base + off_i,
                                   base + off_t)
        dist.all_gather_into_tensor(gathered, mine)
        torch.cuda.synchronize()
        _native.check(_native.lib().vi_merge_partials_packed_device(local_rank, nq, k, world, gathered.data_ptr(),
                                                                    Dm.data_ptr(), Im.data_ptr()))
        return Im

    # ---- operating point: nprobe sweep against exact ground truth -----------------------------
    gt = ground_truth(xb, xq, k)
    sweep = {}
    chosen = args.nprobe
    for p in ([] if chosen else [1, 2, 4, 8, 16, 32, 64]):
        r1, ri = recalls(step(p), gt)
        sweep[p] = {"recall_1nn_at_k": round(r1, 4), "recall_at_k": round(ri, 4)}
        if not chosen and ri >= 0.95:
            chosen = p
            break
    if not chosen:
        chosen = 64
    if chosen not in sweep:
        r1, ri = recalls(step(chosen), gt)
        sweep[chosen] = {"recall_1nn_at_k": round(r1, 4), "recall_at_k": round(ri, 4)}

    # ---- timed region -------------------------------------------------------------------------
    for _ in range(args.warmup):
        step(chosen)
    scan_ms, coarse_ms, tot_ms = [], [], []
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(chosen)
        st = index.last_stats()
        scan_ms.append(st["ms_scan"]); coarse_ms.append(st["ms_coarse"]); tot_ms.append(st["ms_total"])
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    st = index.last_stats()
    ms_per_step = elapsed * 1000.0 / args.steps
    qps = nq * args.steps / elapsed

    # BASELINE configs[1] names nprobe=32 as the headline setting: the same batch there, for reference
    at32 = None
    if chosen != 32 and not args.nprobe:
        r1_32, ri_32 = recalls(step(32), gt)
        for _ in range(2):
            step(32)
        barrier()
        t32 = time.perf_counter()
        for _ in range(5):
            step(32)
        barrier()
        e32 = time.perf_counter() - t32
        if world > 1:
            t = torch.tensor([e32], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            e32 = float(t.item())
        at32 = {"queries_per_s": round(nq * 5 / e32, 1), "ms_per_step": round(e32 * 1000.0 / 5, 4),
                "recall_at_k": round(ri_32, 4), "recall_1nn_at_k": round(r1_32, 4)}
        step(chosen)  # the statistics read below belong to the headline setting
        st = index.last_stats()

    # ---- roofline of the dominant kernel (list scan), from HIP events on the library's stream ----
    scanned = torch.tensor([st["scanned_vectors"]], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(scanned)  # Σ over ranks = what a single GPU would scan
    algo_bytes_rank = st["scanned_vectors"] * (4 * args.d + 8)  # 4·D per vector + 8 B id
    scan_s = float(np.mean(scan_ms)) / 1000.0
    algo_gbs = algo_bytes_rank / scan_s / 1e9 if scan_s > 0 else 0.0
    common = {"algorithmic_bytes_per_launch": int(algo_bytes_rank), "avg_launch_ms": round(scan_s * 1000, 4),
              "coarse_ms": round(float(np.mean(coarse_ms)), 4), "pipeline_ms": round(float(np.mean(tot_ms)), 4)}
    if st["filter_tile_blocks"] > 0:
        # MFMA path.  The list scan is a dense contraction (queries x list vectors x dims): one 64-vector block image
        # staged in LDS is ranked against a group of <= 128 queries on the matrix cores.  Algorithmic flops = 2*D per
        # (query, scanned vector) pair (multiply-add of the norm-expanded distance; SURVEY 8d's 3*D counts the
        # reference's sub/mul/add, which this form does not execute).
        tiles = st["filter_tile_blocks"]
        dq = 4 * ((args.d + 15) // 16)
        mode = int(st["rank_mode"])  # 1 f32 MFMA, 2 bf16x3, 3 bf16 on hi planes only
        flops = 2.0 * args.d * st["scanned_vectors"]
        tf = flops / scan_s / 1e12 if scan_s > 0 else 0.0
        image = 64 * dq * (8 if mode == 3 else 16)       # bytes of one block image streamed per tile
        gq = int(st["group_queries"]) or 128
        tile_bytes = tiles * (image + 2 * gq * 16)        # + 2 x gq block records of 16 B per tile
        stream = tile_bytes / scan_s / 1e9 if scan_s > 0 else 0.0
        extra = {"queries_per_work_item": gq, "flops_per_launch": flops, "useful_TFLOPs": round(tf, 1), "tiles_per_launch": int(tiles),
                 "tile_bytes_per_launch": int(tile_bytes), "tile_stream_GBps": round(stream, 1),
                 "survey_accounting_GBps": round(algo_gbs, 1), "hbm_peak_GBps": HBM_PEAK_GBS, **common}
        if mode == 3:
            # stored values are bf16-exact (8-bit descriptors): ONE bf16 MFMA per 16 dims, the matrix pipe is at
            # ~15 % — what limits the kernel is the stream of tiles (hi image in, block records out) it must move.
            roofline = {"kernel": "filter_kernel<NG,1,false,2> (bf16 MFMA ranking, hi planes only, of query-group x "
                                  "list-segment tiles)", "bound": "hbm", "achieved": round(stream, 1),
                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(stream / HBM_PEAK_GBS, 4), "traffic": None,
                        "mfma_frac_of_bf16_peak": round(tf / MFMA_BF16_PEAK_TF, 4), **extra,
                        "note": "achieved = bytes the kernel's decomposition must move per launch (one hi-plane block "
                                "image per 128-query group + its block records) / launch time; traffic = what PMC saw "
                                "cross to HBM/MALL (the query groups of a list re-read its blocks from L2/MALL); SURVEY "
                                "8d's no-reuse accounting (4*D+8 B per (query, scanned vector)) is "
                                "survey_accounting_GBps"}
        else:
            peak = MFMA_BF16_PEAK_TF / 3.0 if mode == 2 else MFMA_F32_PEAK_TF
            roofline = {"kernel": ("filter_kernel<NG,1,false,1> (bf16x3 MFMA ranking" if mode == 2 else
                                   "filter_kernel<NG,1,false,0> (f32 MFMA ranking") + " of query-group x list-segment tiles)",
                        "bound": "mfma", "achieved": round(tf, 1), "peak": round(peak, 1), "unit": "TFLOP/s",
                        "frac": round(tf / peak, 4), "traffic": None,
                        "peak_note": "bf16 dense MFMA peak 2516 TFLOP/s / 3 split products per multiply" if mode == 2
                                     else "f32 dense MFMA peak", **extra}
    else:
        roofline = {"kernel": "scan_kernel<LISTS> (inverted-list L2 scan + wave top-k)", "bound": "hbm",
                    "achieved": round(algo_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(algo_gbs / HBM_PEAK_GBS, 4), "traffic": None, **common,
                    "note": "achieved counts 4*D+8 B per (query, scanned vector); a list block loaded once is reused "
                            "by up to 8 queries from registers, so the algorithmic rate can exceed what crosses HBM"}

    # HBM traffic of that kernel from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE of
    # this same command; gfx950 correction applied as MI355X_MICROARCH.md prescribes) when they match this workload
    try:
        with open(os.path.join(ROOT, "profiles", "r01_scan_traffic.json")) as f:
            tr = json.load(f)
        import re
        m = re.search(r"filter_kernel<\d+, \d+, false, (\d+)>", tr["kernel"])
        same_kernel = (int(m.group(1)) + 1 == int(st["rank_mode"])) if m else (st["filter_tile_blocks"] == 0)
        if tr["workload"] == [args.n, args.d, args.nlist, chosen, nq, k] and world == 1 and same_kernel:
            roofline["traffic"] = tr["hbm_bytes_per_launch"]
            roofline["traffic_source"] = tr["source"]
    except Exception:
        pass

    # ---- second BASELINE metric: k-means assign on the matrix cores (config C3) ---------------------------
    kmeans = None
    if rank == 0 and world == 1 and not args.no_kmeans:
        import ctypes as C
        del xb
        torch.cuda.empty_cache()
        n3, d3, k3 = args.assign_n, 128, 16384
        g = torch.Generator(device=device); g.manual_seed(42)
        X3 = torch.randn(n3, d3, generator=g, device=device)
        C3 = X3[torch.randperm(n3, generator=g, device=device)[:k3]].contiguous()
        lab = torch.empty(n3, dtype=torch.int32, device=device)
        torch.cuda.synchronize()
        ast = _native.AssignStats()
        best = None
        for _ in range(2):
            _native.check(_native.lib().vi_assign_device(local_rank, X3.data_ptr(), n3, d3, C3.data_ptr(), k3, 42, 1,
                                                         lab.data_ptr(), C.byref(ast)))
            if best is None or ast.ms_total < best[0]:
                best = (ast.ms_total, ast.ms_filter, int(ast.ambiguous_rows))
        flops = 2.0 * n3 * k3 * d3
        tf = flops / (best[1] * 1e-3) / 1e12
        kmeans = {"workload": f"exact nearest-centroid assign N={n3} D={d3} k={k3} (BASELINE config C3), one full pass",
                  "ms_total": round(best[0], 2), "ms_mfma_filter": round(best[1], 2), "ambiguous_rows_rechecked": best[2],
                  "roofline": ({"kernel": "mfma_assign_bf16_kernel<16> (v_mfma_f32_32x32x16_bf16, operands split hi+lo: 3 "
                                          "products per multiply)", "bound": "mfma", "achieved": round(tf, 1),
                                "peak": round(MFMA_BF16_PEAK_TF / 3.0, 1), "unit": "TFLOP/s",
                                "frac": round(tf / (MFMA_BF16_PEAK_TF / 3.0), 4), "flops_per_launch": flops,
                                "peak_note": "bf16 dense MFMA peak 2516 TFLOP/s / 3 split products per multiply"}
                               if os.environ.get("VI_ASSIGN_BF16", "1") != "0" else
                               {"kernel": "mfma_assign_kernel<16,1> (v_mfma_f32_32x32x2_f32)", "bound": "mfma",
                                "achieved": round(tf, 1), "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s",
                                "frac": round(tf / MFMA_F32_PEAK_TF, 4), "flops_per_launch": flops}),
                  "hbm_GBps": round((4.0 * n3 * d3 + 4.0 * n3) / (best[1] * 1e-3) / 1e9, 1),
                  "note": "labels are bit-identical to assign_points_brute_force: rows whose MFMA margin is not "
                          "provably safe are re-evaluated in the reference's exact summation order"}
        del X3, C3, lab

    # ---- CPU baseline: the oracle (C restatement of the reference's CPU path) on the host cores -------
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import oracle_lib as O
        orc = O.OracleIndex.load(os.path.join(work, "index"), os.path.join(work, "shards"))
        threads = min(O.lib().orc_max_threads(), O.usable_cpus())
        xq_h = xq.cpu().numpy()
        done, t0c, chunk = 0, time.perf_counter(), 500
        while time.perf_counter() - t0c < args.cpu_seconds and done < nq:
            m = min(chunk, nq - done)
            rc, Do, Io = orc.search_batch(xq_h[done:done + m], k, chosen, threads)
            if done == 0:  # parity spot check on the same queries
                assert (Io == step(chosen)[:m].cpu().numpy()).all(), "GPU ids differ from the CPU oracle"
            done += m
        cpu_s = time.perf_counter() - t0c
        cpu = {"value": round(done / cpu_s, 1), "unit": "queries/s", "cores": int(threads), "kind": "port",
               "sample": f"{done} of the {nq} bench queries, same index files, nprobe={chosen}, k={k}, lists preloaded "
                         f"in RAM (the reference additionally re-reads shard files per query), host has "
                         f"{os.cpu_count()} logical cores"}

    if rank == 0:
        out = {"metric": "QPS at recall@10>=0.95 (IVF search, SIFT1M-shaped)", "value": round(qps, 1),
               "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
               "scaling": "strong" if world > 1 else "weak", "vs_baseline": None, "dtype": "f32",
               "data": "synthetic (SIFT1M-shaped mixture, seed 42; SIFT1M itself is not available offline)",
               "config": {"workload": f"IVF search N={args.n} D={args.d} nlist={args.nlist} nprobe={chosen} k={k} "
                                      f"nq/step={nq}", "nprobe": chosen, "recall": sweep[chosen],
                          "nprobe_sweep": sweep, "at_nprobe_32": at32, "index_centroids": index.num_centroids,
                          "build_s": round(build_s, 1),
                          "parallelism": f"every list striped over {world} GPU(s) (block b on rank b % N), coarse table replicated"
                                         + (", RCCL all-gather of per-rank top-k" if world > 1 else "")},
               "roofline": roofline, "cpu_baseline": cpu, "kmeans_assign": kmeans}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
