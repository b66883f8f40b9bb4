#!/usr/bin/env python3
"""kt_timeline.py kernel_trace.csv: the kernels of the LAST search in a rocprofv3 kernel trace (from split_queries_kernel to
select_kernel) with their durations and the gaps between them, in microseconds."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ends = [i for i, r in enumerate(rows) if "select_kernel" in r["Kernel_Name"] and "coarse" not in r["Kernel_Name"]]
last = ends[-1]
first = max(i for i in range(last) if "split_queries_kernel" in rows[i]["Kernel_Name"])
t0 = int(rows[first]["Start_Timestamp"])
prev_end = t0
busy = 0
for r in rows[first:last + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("vi::(anonymous namespace)::", "").replace("void ", "")
    print("%8.1f  +%5.1f gap  %7.1f us  %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, name[:100]))
    busy += e - s
    prev_end = e
print("total %.1f us, kernels %.1f us, gaps %.1f us" % ((prev_end - t0) / 1e3, busy / 1e3, (prev_end - t0 - busy) / 1e3))
