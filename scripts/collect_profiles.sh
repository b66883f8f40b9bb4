#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root: rocprofv3 kernel trace + the two PMC passes of the
# bench command, written under gpurun_out/prof_<tag>/ ; scripts/summarize_profiles.py turns them into
# the files committed under profiles/.
set -u
TAG=${1:-r02}
NPROBE=${2:-32}
R=$PWD
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/bench.py --steps 5 --warmup 1 --nprobe $NPROBE --no-cpu-baseline --no-kmeans --no-extras"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_kt -- $CMD > $R/gpurun_out/prof_${TAG}_kt.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_${TAG}_fetch -- $CMD > $R/gpurun_out/prof_${TAG}_fetch.log 2>&1 || exit 2
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_${TAG}_write -- $CMD > $R/gpurun_out/prof_${TAG}_write.log 2>&1 || exit 3
echo profiles collected
