#!/bin/bash
# PMC passes over the SMPL-only bench (separate rocprofv3 runs, kernel-trace only: gpurun refuses --pmc with the wider traces).
# usage: TAG=name [BENCH="python tools/with_tools_lib.py bench.py"] [GROUPS_SEL="1 2 7 8 9"] bash tools/pmc_smpl_w.sh [bench flags, e.g. --smpl-kernel wide]
R=$PWD; export TMPDIR=/tmp; TAG=${TAG:-pmc_smpl_w}; mkdir -p $R/gpurun_out/$TAG
BENCH=${BENCH:-"python $R/bench.py"}
cd /tmp
i=0
for C in "MfmaUtil VALUBusy" "GRBM_GUI_ACTIVE GRBM_COUNT" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_MFMA" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS"; do
  i=$((i+1))
  if [ -n "$GROUPS_SEL" ] && ! echo " $GROUPS_SEL " | grep -q " $i "; then continue; fi
  timeout 120 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/$TAG/p$i -- $BENCH --workload smpl --steps 2 --warmup 1 --no-cpu-baseline --no-graph --no-reduced-ab "$@" > $R/gpurun_out/$TAG/p$i.log 2>&1
  f=$(find $R/gpurun_out/$TAG/p$i -name '*counter_collection.csv' | head -1)
  echo "== $C"; [ -n "$f" ] && python $R/tools/pmc_summary.py $f | grep "smpl_verts" || tail -2 $R/gpurun_out/$TAG/p$i.log
done > $R/gpurun_out/$TAG/summary.txt 2>&1
cat $R/gpurun_out/$TAG/summary.txt
rm -rf $R/gpurun_out/$TAG/p*/
