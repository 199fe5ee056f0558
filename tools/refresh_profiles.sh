#!/bin/bash
# Re-create everything under profiles/ (run on the GPU box through gpurun; results land in gpurun_out/refresh)
R=$PWD; export TMPDIR=/tmp; O=$R/gpurun_out/refresh; mkdir -p $O
cd /tmp
prof() {  # tag, bench args...
  tag=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$tag -- python $R/bench.py "$@" > $O/prof_$tag.log 2>&1
  f=$(find $O/prof_$tag -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/r01_${tag}_kernel_stats.csv
}
prof train_b64 --steps 10 --warmup 3 --no-cpu-baseline --no-graph --no-overlap
prof fwd_b64 --workload fwd --steps 10 --warmup 3 --no-cpu-baseline --no-graph
prof smpl_65536 --workload smpl --steps 5 --warmup 2 --no-cpu-baseline --no-graph
i=0
for C in "MfmaUtil" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/pmc$i -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-graph --no-overlap > $O/pmc$i.log 2>&1
done
cp $R/profiles/pmc_traffic.json $O/pmc_traffic.json 2>/dev/null
python $R/tools/pmc_summary.py --json $O/pmc_traffic.json train_r18_b64 'rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python bench.py --steps 2 --warmup 1 --no-graph; tools/refresh_profiles.sh' $(find $O/pmc1 $O/pmc2 $O/pmc3 -name '*counter_collection.csv') > $O/r01_train_b64_pmc_summary.txt 2>&1
i=3
for C in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/pmc$i -- python $R/bench.py --workload fwd --steps 2 --warmup 1 --no-cpu-baseline --no-graph > $O/pmc$i.log 2>&1
done
python $R/tools/pmc_summary.py --json $O/pmc_traffic.json fwd_r18_b64 'rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python bench.py --workload fwd --steps 2 --warmup 1 --no-graph; tools/refresh_profiles.sh' $(find $O/pmc4 $O/pmc5 -name '*counter_collection.csv') > $O/r01_fwd_b64_pmc_summary.txt 2>&1
cp $O/pmc_traffic.json $R/profiles/pmc_traffic.json      # bench.py reports roofline.traffic from this file
cd $R
python bench.py 2>/dev/null | tail -1 > $O/r01_bench_train_b64.json
python bench.py --workload fwd 2>/dev/null | tail -1 > $O/r01_bench_fwd_b64.json
python bench.py --workload smpl --steps 10 --warmup 3 2>/dev/null | tail -1 > $O/r01_bench_smpl_65536.json
python bench.py --layers 50 --batch 32 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r01_bench_train_r50_b32.json
rm -rf $O/prof_*/ $O/pmc*/
ls -la $O
