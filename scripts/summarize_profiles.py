#!/usr/bin/env python3
"""Turn the rocprofv3 outputs of scripts/collect_profiles.sh (gpurun_out/prof_<tag>_*) into the committed summaries
under profiles/:

  <tag>_headline_kernel_stats.csv   kernel-trace stats of scripts/profile_headline.py (every search launch at nprobe 32)
  <tag>_headline_profile.json       its JSON line + per-kernel averages + HBM traffic of the dominant kernel
  <tag>_scan_traffic.json           the traffic record bench.py reads for roofline.traffic (same workload, same kernel)
  <tag>_c3_kernel_stats.csv         kernel-trace stats of scripts/profile_c3.py (C3 assign + update passes)
  <tag>_c3_profile.json             its JSON line + per-kernel averages + FETCH/WRITE bytes of the assign and update kernels

FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half the bytes of a wide coalesced stream
(MI355X_MICROARCH.md, HBM section) and is doubled here."""
import csv
import glob
import json
import os
import re
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"


def newest(pattern):
    """gpurun merges every call's output into gpurun_out/: take the most recent profile"""
    return max(glob.glob(pattern, recursive=True), key=os.path.getmtime)


def json_line(log):
    for line in open(log, errors="replace"):
        if line.startswith('{"workload"'):
            return json.loads(line)
    return None


def counter_avg(name, counter, kernel_re, last=None):
    """mean counter value per launch over the launches of the kernels matching kernel_re"""
    f = newest(f"gpurun_out/prof_{tag}_{name}/**/*_counter_collection.csv")
    per = {}
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter and re.search(kernel_re, r["Kernel_Name"]):
            per.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
    out = {}
    for k, v in per.items():
        v = v[-last:] if last else v
        out[k] = (sum(v) / len(v), len(v))
    return out


def kernel_stats(name, dst):
    f = newest(f"gpurun_out/prof_{tag}_{name}/**/*_kernel_stats.csv")
    shutil.copy(f, dst)
    return {r["Name"]: r for r in csv.DictReader(open(f))}


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return re.sub(r"\(.*$", "", n)


# ---- headline -------------------------------------------------------------------------------------------------------
SEARCH = r"rank_stream_kernel|filter_kernel<|select_kernel|coarse_select|group_scan|group_prepare|group_scatter|query_offsets|list_totals|cursor_kernel|item_desc|item_cols|split_queries|scan_kernel<\d+, 0"
RANK = r"rank_stream_kernel<|filter_kernel<\d+, \d+, false"
ks = kernel_stats("head_kt", f"profiles/{tag}_headline_kernel_stats.csv")
line = json_line(f"gpurun_out/prof_{tag}_head_kt.log")
kern = {short(n): {"calls": int(r["Calls"]), "avg_us": round(float(r["AverageNs"]) / 1e3, 2), "min_us": round(float(r["MinNs"]) / 1e3, 2),
                   "max_us": round(float(r["MaxNs"]) / 1e3, 2)} for n, r in ks.items() if re.search(SEARCH, n)}
fetch = counter_avg("head_fetch", "FETCH_SIZE", RANK)
write = counter_avg("head_write", "WRITE_SIZE", RANK)
rank_name = max(fetch, key=lambda k: fetch[k][1])
fk, wk = fetch[rank_name][0], write[rank_name][0]
rank_row = ks[rank_name]
m = re.search(r"filter_kernel<\d+, \d+, false, (\d+), \d+>", rank_name) or re.search(r"rank_stream_kernel<\d+, (\d+), ", rank_name)
traffic = {"workload": line["workload"], "kernel": rank_name, "rank_mode": line["rank_mode"], "fetch_size_kib_avg": fk, "write_size_kib_avg": wk,
           "hbm_bytes_per_launch": int((2.0 * fk + wk) * 1024), "avg_launch_ns_rocprof": float(rank_row["AverageNs"]),
           "calls": int(rank_row["Calls"]),
           "source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of `scripts/profile_headline.py` (warm-up + timed steps at "
                     f"nprobe {line['workload'][3]} only: every launch is the headline operating point); bytes = (2*FETCH_SIZE + WRITE_SIZE) KiB "
                     f"(gfx950 FETCH_SIZE half-count correction), mean of all {fetch[rank_name][1]} launches; profiles/{tag}_headline_kernel_stats.csv "
                     f"holds the kernel-trace stats of the same program"}
json.dump(traffic, open(f"profiles/{tag}_scan_traffic.json", "w"), indent=1)
flops = 2.0 * line["workload"][1] * line["scanned_vectors"]
json.dump({"program": "scripts/profile_headline.py --steps 10 --warmup 2", "line": line, "search_kernels": kern,
           "dominant_kernel": {"name": rank_name, "avg_launch_us": round(float(rank_row["AverageNs"]) / 1e3, 2),
                               "useful_flop_per_launch": flops,
                               "useful_TFLOPs": round(flops / float(rank_row["AverageNs"]) / 1e3, 1),
                               "hbm_bytes_per_launch": traffic["hbm_bytes_per_launch"],
                               "hbm_TBps": round(traffic["hbm_bytes_per_launch"] / float(rank_row["AverageNs"]) / 1e3, 3)}},
          open(f"profiles/{tag}_headline_profile.json", "w"), indent=1)
print(json.dumps(traffic, indent=1))

# ---- C3 -------------------------------------------------------------------------------------------------------------
C3 = r"vi::.*(mfma_assign|scan_kernel<4, 1|segment_|radix_|offsets_kernel|label_range|gather_amb|centroid_)"
ks = kernel_stats("c3_kt", f"profiles/{tag}_c3_kernel_stats.csv")
line = json_line(f"gpurun_out/prof_{tag}_c3_kt.log")
fetch = counter_avg("c3_fetch", "FETCH_SIZE", C3)
write = counter_avg("c3_write", "WRITE_SIZE", C3)
kern = {}
for n, r in ks.items():
    if re.search(C3, n):
        e = {"calls": int(r["Calls"]), "avg_us": round(float(r["AverageNs"]) / 1e3, 2)}
        if n in fetch and n in write:
            b = (2.0 * fetch[n][0] + write[n][0]) * 1024
            e["hbm_bytes_per_launch"] = int(b)
            e["hbm_TBps"] = round(b / float(r["AverageNs"]) / 1e3, 3)
        kern[short(n)] = e
json.dump({"program": "scripts/profile_c3.py --passes 3 --warmup 1", "line": line, "kernels": kern,
           "note": "hbm_bytes_per_launch = (2*FETCH_SIZE + WRITE_SIZE) KiB from separate --pmc passes (gfx950 correction), mean over "
                   "the kernel's launches"}, open(f"profiles/{tag}_c3_profile.json", "w"), indent=1)
print(json.dumps(kern, indent=1))
