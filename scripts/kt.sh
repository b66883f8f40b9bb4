#!/bin/bash
# Run ON THE GPU BOX: rocprofv3 kernel trace + stats of one of this repo's python scripts (the program itself after `--`).
#   scripts/kt.sh <tag> <script.py> [args...]   -> gpurun_out/kt_<tag>/ , top kernels printed
set -u
TAG=$1; shift
R=$PWD
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/kt_$TAG
S=$1; shift
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kt_$TAG -- python3 $R/$S "$@" > $R/gpurun_out/kt_$TAG.log 2>&1 || { tail -20 $R/gpurun_out/kt_$TAG.log; exit 1; }
tail -3 $R/gpurun_out/kt_$TAG.log
python3 - <<PY
import csv, glob
f = glob.glob("$R/gpurun_out/kt_$TAG/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:24]:
    print("%-100s calls %5s avg_us %10.1f total_ms %9.2f" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
