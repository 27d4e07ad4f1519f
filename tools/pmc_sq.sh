#!/bin/bash
# SQ counter passes over the serialised bench (solo kernels): gpurun_out/pmc_sq_summary.txt
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export LVX_SERIAL=1
rm -rf gpurun_out/pmc_sq1 gpurun_out/pmc_sq2
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d gpurun_out/pmc_sq1 -o pmc -- python bench.py --steps 3 --warmup 1 --no-secondary --no-cpu-baseline > gpurun_out/pmc_sq1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_sq2 -o pmc -- python bench.py --steps 3 --warmup 1 --no-secondary --no-cpu-baseline > gpurun_out/pmc_sq2.log 2>&1
python tools/pmc_summary.py gpurun_out/pmc_sq1 gpurun_out/pmc_sq2 > gpurun_out/pmc_sq_summary.txt
tail -2 gpurun_out/pmc_sq2.log
