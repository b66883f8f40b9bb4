#!/bin/bash
# Run ON THE GPU BOX: kernel trace of the search pipeline on the bench workload (scripts/gpu_rank_exp.py, SET / nprobe from the environment)
set -u
R=$PWD
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/kt_rank
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kt_rank -- python3 $R/scripts/gpu_rank_exp.py ${NPROBE:-32} > $R/gpurun_out/kt_rank.log 2>&1 || exit 1
python3 - <<PY
import csv, glob
f = glob.glob("$R/gpurun_out/kt_rank/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:28]:
    print("%-110s calls %5s avg_us %9.1f total_ms %8.2f" % (r["Name"][:110], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
