#!/bin/bash
# Run ON THE GPU BOX: SQ / TA / TCP counters of the two select kernels at the headline point (scripts/profile_headline.py),
# one --pmc pass per counter group.  Usage: scripts/pmc_select.sh [tag]   (environment knobs are inherited)
set -u
R=$PWD
TAG=${1:-sel}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/scripts/profile_headline.py --steps 4 --warmup 1"
declare -A G
G[a]="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU"
G[b]="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_SMEM"
G[c]="TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum"
G[d]="TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TOTAL_READ_sum SQ_IFETCH_LEVEL SQC_ICACHE_MISSES SQC_ICACHE_REQ SQ_INSTS_BRANCH SQ_WAVES"
for g in ${GROUPS_TO_RUN:-a b}; do   # (c d: TA / TCP / instruction-cache counters — a pass takes ~3 minutes: the index build runs under the counters too)
  rm -rf $R/gpurun_out/pmc_${TAG}_$g
  rocprofv3 --pmc ${G[$g]} --output-format csv -d $R/gpurun_out/pmc_${TAG}_$g -- $CMD > $R/gpurun_out/pmc_${TAG}_$g.log 2>&1 || { echo "pass $g failed"; tail -5 $R/gpurun_out/pmc_${TAG}_$g.log; }
done
python3 - <<PY
import csv, glob, collections
for g in "abcd":
    fs = glob.glob("$R/gpurun_out/pmc_${TAG}_%s/**/*counter_collection.csv" % g, recursive=True)
    if not fs:
        continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(set)
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"]
        hit = [x for x in ("select_kernel<", "coarse_select_direct") if x in k]
        if hit:
            acc[hit[0]][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[hit[0]].add(r["Dispatch_Id"])
    for k, v in sorted(acc.items()):
        for c, x in sorted(v.items()):
            print("%-22s %-36s %.5g per launch" % (k, c, x / len(cnt[k])))
PY
