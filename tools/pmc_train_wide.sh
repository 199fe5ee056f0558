#!/bin/bash
# wide PMC sweep over the eager training step (separate runs, kernel-trace only); prints the MFMA kernels' rows
R=$PWD; export TMPDIR=/tmp; TAG=${TAG:-pmc_train_wide}; mkdir -p $R/gpurun_out/$TAG
cd /tmp
i=0
for C in "MfmaUtil VALUBusy" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAVES SQ_LDS_IDX_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_LDS_DMA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/$TAG/p$i -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-graph --no-overlap --no-stem-ab "$@" > $R/gpurun_out/$TAG/p$i.log 2>&1
  f=$(find $R/gpurun_out/$TAG/p$i -name '*counter_collection.csv' | head -1)
  echo "== $C"; [ -n "$f" ] && python $R/tools/pmc_summary.py $f | grep -E "conv_|stem_" || tail -2 $R/gpurun_out/$TAG/p$i.log
done > $R/gpurun_out/$TAG/summary.txt 2>&1
cat $R/gpurun_out/$TAG/summary.txt
rm -rf $R/gpurun_out/$TAG/p*/
