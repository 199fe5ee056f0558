#!/bin/bash
# wider PMC sweep over the SMPL-only bench (separate runs, kernel-trace only).  args: extra bench flags
R=$PWD; export TMPDIR=/tmp; TAG=${TAG:-pmc_smpl2}; mkdir -p $R/gpurun_out/$TAG
cd /tmp
i=0
for C in "MfmaUtil VALUBusy" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM" "GRBM_GUI_ACTIVE GRBM_COUNT" "SQ_IFETCH SQ_IFETCH_LEVEL" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS"; do
  i=$((i+1))
  timeout 120 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/$TAG/p$i -- python $R/bench.py --workload smpl --steps 2 --warmup 1 --no-cpu-baseline --no-graph "$@" > $R/gpurun_out/$TAG/p$i.log 2>&1
  f=$(find $R/gpurun_out/$TAG/p$i -name '*counter_collection.csv' | head -1)
  echo "== $C"; [ -n "$f" ] && python $R/tools/pmc_summary.py $f | grep "smpl_verts" || tail -2 $R/gpurun_out/$TAG/p$i.log
done > $R/gpurun_out/$TAG/summary.txt 2>&1
cat $R/gpurun_out/$TAG/summary.txt
rm -rf $R/gpurun_out/$TAG/p*/
