#!/bin/bash
# PMC passes over one implicit-GEMM shape (tools/igemm_one.py), separate runs, kernel-trace only
R=$PWD; export TMPDIR=/tmp; mkdir -p $R/gpurun_out/pmc_igemm
cd /tmp
i=0
for C in "MfmaUtil VALUBusy" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmc_igemm/p$i -- python $R/tools/igemm_one.py "$@" > $R/gpurun_out/pmc_igemm/p$i.log 2>&1
  f=$(find $R/gpurun_out/pmc_igemm/p$i -name '*counter_collection.csv' | head -1)
  echo "== $C"; [ -n "$f" ] && python $R/tools/pmc_summary.py $f | grep "igemm" || tail -2 $R/gpurun_out/pmc_igemm/p$i.log
done
