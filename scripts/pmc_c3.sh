#!/bin/bash
# Run ON THE GPU BOX: SQ counters of the C3 assign kernels (scripts/profile_c3.py, one pass), separate --pmc passes.
set -u
R=$PWD
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/scripts/profile_c3.py --passes 1 --warmup 0"
rm -rf $R/gpurun_out/pmc_c3_a $R/gpurun_out/pmc_c3_b
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $R/gpurun_out/pmc_c3_a -- $CMD > $R/gpurun_out/pmc_c3_a.log 2>&1 || exit 1
rocprofv3 --pmc SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS --output-format csv -d $R/gpurun_out/pmc_c3_b -- $CMD > $R/gpurun_out/pmc_c3_b.log 2>&1 || exit 2
python3 - <<PY
import csv, glob, collections
for d in ("pmc_c3_a", "pmc_c3_b"):
    f = glob.glob("$R/gpurun_out/%s/**/*counter_collection.csv" % d, recursive=True)[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.Counter()
    for r in csv.DictReader(open(f)):
        if "mfma_assign" in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:70]][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, v in acc.items():
        print(k)
        for c, x in sorted(v.items()):
            print("   %-28s %.4g" % (c, x))
PY
