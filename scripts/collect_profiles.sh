#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root.  rocprofv3 of the two BASELINE metrics, each from a program that
# runs nothing but that operating point (scripts/profile_headline.py: C2 at nprobe 32; scripts/profile_c3.py: the C3
# assign + update passes): a kernel trace with stats, then FETCH_SIZE and WRITE_SIZE in separate --pmc passes (never
# combined with a trace domain).  Output under gpurun_out/prof_<tag>_*; scripts/summarize_profiles.py turns it into the
# files committed under profiles/.
set -u
TAG=${1:-r03}
R=$PWD
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
run() {  # name, rocprof args..., -- program
  local name=$1; shift
  rm -rf $R/gpurun_out/prof_${TAG}_$name
  rocprofv3 "$@" > $R/gpurun_out/prof_${TAG}_$name.log 2>&1 || { echo "FAILED: $name"; tail -5 $R/gpurun_out/prof_${TAG}_$name.log; exit 1; }
  echo "done: $name"
}
H="python3 $R/scripts/profile_headline.py --steps 10 --warmup 2"
K="python3 $R/scripts/profile_c3.py --passes 3 --warmup 1"
run head_kt --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_head_kt -- $H
run head_fetch --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_${TAG}_head_fetch -- $H
run head_write --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_${TAG}_head_write -- $H
run c3_kt --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_c3_kt -- $K
run c3_fetch --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_${TAG}_c3_fetch -- $K
run c3_write --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_${TAG}_c3_write -- $K
if [ "${3:-}" = "x3" ]; then
  H3="$H --real-valued"
  run head3_kt --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_head3_kt -- $H3
fi
grep -h workload $R/gpurun_out/prof_${TAG}_head_kt.log $R/gpurun_out/prof_${TAG}_c3_kt.log
echo profiles collected
