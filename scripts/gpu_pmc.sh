#!/bin/bash
mkdir -p gpurun_out/pmc
export TMPDIR=/tmp KB_ITERS=2
cd /tmp
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc -o sq -- python $GRAFT_REPO_ROOT/scripts/kernel_bench.py 256 > $GRAFT_REPO_ROOT/gpurun_out/pmc/sq.log 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc -o inst -- python $GRAFT_REPO_ROOT/scripts/kernel_bench.py 256 > $GRAFT_REPO_ROOT/gpurun_out/pmc/inst.log 2>&1
timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TA_BUSY_avr --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc -o tcc -- python $GRAFT_REPO_ROOT/scripts/kernel_bench.py 256 > $GRAFT_REPO_ROOT/gpurun_out/pmc/tcc.log 2>&1
cd $GRAFT_REPO_ROOT
ls gpurun_out/pmc | head -20; tail -3 gpurun_out/pmc/tcc.log
