#!/bin/bash
# PMC passes over the SMPL-only bench (separate runs, kernel-trace only -- never with sys/hip traces)
R=$PWD; export TMPDIR=/tmp; mkdir -p $R/gpurun_out/pmc
cd /tmp
i=0
for C in "MfmaUtil VALUBusy" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM" "FETCH_SIZE WRITE_SIZE" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  timeout 240 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmc/p$i -- python $R/bench.py --workload smpl --steps 2 --warmup 1 --no-cpu-baseline --no-graph > $R/gpurun_out/pmc/p$i.log 2>&1
  f=$(find $R/gpurun_out/pmc/p$i -name '*counter_collection.csv' | head -1)
  echo "== $C"; [ -n "$f" ] && python $R/tools/pmc_summary.py $f | grep smpl || tail -3 $R/gpurun_out/pmc/p$i.log
done
