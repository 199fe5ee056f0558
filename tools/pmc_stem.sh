#!/bin/bash
# PMC passes over the stem weight-gradient sweep (separate runs, kernel-trace only)
R=$PWD; export TMPDIR=/tmp; mkdir -p $R/gpurun_out/pmc_stem
cd /tmp
i=0
for C in "MfmaUtil VALUBusy" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmc_stem/p$i -- python $R/tools/stem_wgrad_sweep.py > $R/gpurun_out/pmc_stem/p$i.log 2>&1
  f=$(find $R/gpurun_out/pmc_stem/p$i -name '*counter_collection.csv' | head -1)
  echo "== $C"; [ -n "$f" ] && python $R/tools/pmc_summary.py $f | grep "stem_wgrad_kernel" || tail -3 $R/gpurun_out/pmc_stem/p$i.log
done
