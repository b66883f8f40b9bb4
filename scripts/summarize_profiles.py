#!/usr/bin/env python3
"""Turn the rocprofv3 outputs of scripts/collect_profiles.sh (gpurun_out/prof_<tag>_*) into the committed
summaries under profiles/: kernel stats CSV and the HBM traffic of the list-scan kernel.
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half the bytes of a wide coalesced stream
(MI355X_MICROARCH.md, HBM section) and is doubled here."""
import csv
import re
import glob
import json
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
workload = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1_000_000, 128, 4096, 32, 10_000, 10]
import os


def newest(pattern):
    """gpurun merges every call's output into gpurun_out/: take the most recent profile"""
    return max(glob.glob(pattern), key=os.path.getmtime)


kt = newest(f"gpurun_out/prof_{tag}_kt/*/*_kernel_stats.csv")
shutil.copy(kt, f"profiles/{tag}_bench_kernel_stats.csv")


# dominant kernel of the pipeline: the MFMA list ranking when the MFMA path ran, else the VALU list scan
KERNEL = r"rank_stream_kernel<\d+, \d+, (false|true), \d+>|filter_kernel<\d+, \d+, false, \d+, \d+>|scan_kernel<\d+, 0, false, false>"


def scan_avg(pattern, counter):
    f = newest(pattern)
    rows = [r for r in csv.DictReader(open(f)) if re.search(KERNEL, r["Kernel_Name"])
            and r["Counter_Name"] == counter]
    rows = [r for r in rows if r["Kernel_Name"] == rows[-1]["Kernel_Name"]]  # the timed steps' instantiation
    vals = [float(r["Counter_Value"]) for r in rows]
    vals = vals[-5:]  # the timed steps (same nprobe); earlier launches are warm-up / recall evaluation
    return sum(vals) / len(vals), len(vals), rows[-1]["Kernel_Name"]


fetch, n1, name = scan_avg(f"gpurun_out/prof_{tag}_fetch/*/*_counter_collection.csv", "FETCH_SIZE")
write, n2, _ = scan_avg(f"gpurun_out/prof_{tag}_write/*/*_counter_collection.csv", "WRITE_SIZE")
stats = {r["Name"]: r for r in csv.DictReader(open(kt))}
scan = stats[name]
m = re.search(r"filter_kernel<\d+, \d+, false, (\d+), \d+>", name) or re.search(r"rank_stream_kernel<\d+, (\d+), ", name)
out = {"workload": workload, "kernel": name, "rank_mode": (int(m.group(1)) + 1) if m else 0, "fetch_size_kib_avg": fetch, "write_size_kib_avg": write,
       "hbm_bytes_per_launch": int((2.0 * fetch + write) * 1024),
       "avg_launch_ns_rocprof": float(scan["AverageNs"]), "calls": int(scan["Calls"]),
       "source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of `bench.py --nprobe {workload[3]}`; "
                 f"bytes = (2*FETCH_SIZE + WRITE_SIZE) KiB (gfx950 FETCH_SIZE half-count correction), mean of the last "
                 f"{n1} list-scan launches; profiles/{tag}_bench_kernel_stats.csv holds the kernel-trace stats"}
json.dump(out, open(f"profiles/{tag}_scan_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
