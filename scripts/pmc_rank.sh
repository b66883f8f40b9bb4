#!/bin/bash
# Run ON THE GPU BOX: SQ / TCC counters of the default rank kernel on the bench workload (separate passes).
set -u
R=$PWD
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
export EXP=default
CMD="python3 $R/scripts/gpu_rank_exp.py 16"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $R/gpurun_out/pmc_sq -- $CMD > $R/gpurun_out/pmc_sq.log 2>&1 || exit 1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum --output-format csv -d $R/gpurun_out/pmc_tcc -- $CMD > $R/gpurun_out/pmc_tcc.log 2>&1 || exit 2
rocprofv3 --pmc SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA --output-format csv -d $R/gpurun_out/pmc_sq2 -- $CMD > $R/gpurun_out/pmc_sq2.log 2>&1 || exit 3
echo pmc done
