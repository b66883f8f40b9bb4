#!/bin/bash
# Run ON THE GPU BOX: SQ counters of the search kernels at the headline point (scripts/profile_headline.py), separate --pmc passes.
set -u
R=$PWD
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/scripts/profile_headline.py --steps 4 --warmup 1"
rm -rf $R/gpurun_out/pmc_head_a $R/gpurun_out/pmc_head_b
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $R/gpurun_out/pmc_head_a -- $CMD > $R/gpurun_out/pmc_head_a.log 2>&1 || exit 1
rocprofv3 --pmc SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS --output-format csv -d $R/gpurun_out/pmc_head_b -- $CMD > $R/gpurun_out/pmc_head_b.log 2>&1 || exit 2
python3 - <<PY
import csv, glob, collections
for d in ("pmc_head_a", "pmc_head_b"):
    f = glob.glob("$R/gpurun_out/%s/**/*counter_collection.csv" % d, recursive=True)[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        hit = [x for x in ("rank_stream_kernel", "select_kernel<", "coarse_select_direct", "filter_kernel<16, 1, true") if x in k]
        if hit:
            kk = hit[0]
            acc[kk][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[kk].add(r["Dispatch_Id"])
    for k, v in acc.items():
        print(k, "launches", len(cnt[k]))
        for c, x in sorted(v.items()):
            print("   %-28s %.4g per launch" % (c, x / len(cnt[k])))
PY
