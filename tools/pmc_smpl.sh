#!/bin/bash
# PMC passes over the SMPL-only bench (separate runs, kernel-trace only -- never with sys/hip traces).  args: extra bench flags
R=$PWD; export TMPDIR=/tmp; TAG=${TAG:-pmc_smpl}; mkdir -p $R/gpurun_out/$TAG
cd /tmp
i=0
for C in "MfmaUtil VALUBusy" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES"; do
  i=$((i+1))
  timeout 240 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/$TAG/p$i -- python $R/bench.py --workload smpl --steps 2 --warmup 1 --no-cpu-baseline --no-graph "$@" > $R/gpurun_out/$TAG/p$i.log 2>&1
  f=$(find $R/gpurun_out/$TAG/p$i -name '*counter_collection.csv' | head -1)
  echo "== $C"; [ -n "$f" ] && python $R/tools/pmc_summary.py $f | grep smpl || tail -3 $R/gpurun_out/$TAG/p$i.log
done > $R/gpurun_out/$TAG/summary.txt 2>&1
cat $R/gpurun_out/$TAG/summary.txt
rm -rf $R/gpurun_out/$TAG/p*/
