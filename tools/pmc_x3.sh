#!/bin/bash
# counter passes over the bf16x3 implicit GEMM (one layer shape, a few tile configurations / ablations); kernel-trace only
R=$PWD; export TMPDIR=/tmp; TAG=${TAG:-pmc_x3}; mkdir -p $R/gpurun_out/$TAG
SHAPE=${SHAPE:-l3}; CFGS=${CFGS:-"5 1 133 69 197"}
cd /tmp
for cfg in $CFGS; do
  echo "#### $SHAPE cfg $cfg"
  i=0
  for C in "GRBM_GUI_ACTIVE MfmaUtil" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum"; do
    i=$((i+1))
    timeout 120 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/$TAG/p$i -- python $R/tools/x3_one.py $SHAPE $cfg 6 > $R/gpurun_out/$TAG/p$i.log 2>&1
    f=$(find $R/gpurun_out/$TAG/p$i -name '*counter_collection.csv' | head -1)
    echo "== $C"; [ -n "$f" ] && python $R/tools/pmc_summary.py $f | grep -E "conv_igemm" || tail -2 $R/gpurun_out/$TAG/p$i.log
    rm -rf $R/gpurun_out/$TAG/p$i
  done
done > $R/gpurun_out/$TAG/summary_$SHAPE.txt 2>&1
cat $R/gpurun_out/$TAG/summary_$SHAPE.txt
