#!/bin/bash
# SQ and cache counters of the weight-gradient kernels on the ViT-B linear shapes: register-staged (SAICV_TN_DMA=0) vs LDS-DMA ring
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r03pmctn
mkdir -p $O
for v in 0 1; do
  export SAICV_TN_DMA=$v
  cd /tmp
  timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/sq_$v -o sq -- python $GRAFT_REPO_ROOT/scripts/linear_bench.py > $O/sq_$v.log 2>&1
  timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TA_BUSY_avr --kernel-trace --output-format csv -d $O/tcc_$v -o tcc -- python $GRAFT_REPO_ROOT/scripts/linear_bench.py > $O/tcc_$v.log 2>&1
  cd $GRAFT_REPO_ROOT
  for p in sq tcc; do python scripts/pmc_summarize.py $O/${p}_$v igemm_tn > $O/summary_${p}_dma$v.txt 2>&1; done
  rm -rf $O/*_$v/*/*kernel_trace.csv
done
head -40 $O/summary_sq_dma1.txt
