#!/bin/bash
# SQ / LDS / cache counters of the NT GEMM kernel on the ViT-B linear shapes, KC = 4 vs KC = 8 at the 256 x 256 geometry
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp KB_ITERS=2
O=$GRAFT_REPO_ROOT/gpurun_out/r02pmc
mkdir -p $O
for kc in 4 8; do
  export SAICV_NT_KC=$kc SAICV_NT_TILE=0
  python scripts/linear_bench.py > $O/linear_kc${kc}.log 2>&1
  cat $O/linear_kc${kc}.log | cut -c1-200
  cd /tmp
  timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/sq_kc$kc -o sq -- python $GRAFT_REPO_ROOT/scripts/linear_bench.py > $O/sq_kc$kc.log 2>&1
  timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA --kernel-trace --output-format csv -d $O/inst_kc$kc -o inst -- python $GRAFT_REPO_ROOT/scripts/linear_bench.py > $O/inst_kc$kc.log 2>&1
  timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TA_BUSY_avr --kernel-trace --output-format csv -d $O/tcc_kc$kc -o tcc -- python $GRAFT_REPO_ROOT/scripts/linear_bench.py > $O/tcc_kc$kc.log 2>&1
  timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAIT_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_ACTIVE_INST_FLAT --kernel-trace --output-format csv -d $O/vm_kc$kc -o vm -- python $GRAFT_REPO_ROOT/scripts/linear_bench.py > $O/vm_kc$kc.log 2>&1
  cd $GRAFT_REPO_ROOT
  for p in sq inst tcc vm; do python scripts/pmc_summarize.py $O/${p}_kc$kc igemm_nt > $O/summary_${p}_kc$kc.txt 2>&1; done
  rm -rf $O/*_kc$kc/*/*kernel_trace.csv
done
head -60 $O/summary_sq_kc4.txt
