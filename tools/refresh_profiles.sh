#!/bin/bash
# Re-create everything under profiles/ for round RND (run on the GPU box through gpurun; results land in gpurun_out/refresh, copy the
# summaries into profiles/ afterwards).  Counter passes are separate rocprofv3 runs with --kernel-trace only (never with sys/hip traces).
RND=${RND:-r03}
R=$PWD; export TMPDIR=/tmp; O=$R/gpurun_out/refresh; mkdir -p $O
cd /tmp
prof() {  # tag, bench args...
  tag=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$tag -- python $R/bench.py "$@" > $O/prof_$tag.log 2>&1
  f=$(find $O/prof_$tag -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/${RND}_${tag}_kernel_stats.csv
}
pmc() {  # json tag, summary name, source text, bench args...
  tag=$1; name=$2; src=$3; shift 3
  fs=""
  for C in "MfmaUtil GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
    d=$O/pmc_${name}_$(echo $C | cut -d' ' -f1)
    timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $d -- python $R/bench.py "$@" > $d.log 2>&1
    fs="$fs $(find $d -name '*counter_collection.csv' | head -1)"
  done
  python $R/tools/pmc_summary.py --json $O/pmc_traffic.json $tag "$src" $fs > $O/${RND}_${name}_pmc_summary.txt 2>&1
}
cp $R/profiles/pmc_traffic.json $O/pmc_traffic.json 2>/dev/null
prof train_b64 --steps 10 --warmup 3 --no-cpu-baseline --no-graph --no-overlap --no-stem-ab
prof fwd_b64 --workload fwd --steps 10 --warmup 3 --no-cpu-baseline --no-graph
prof smpl_65536 --workload smpl --steps 5 --warmup 2 --no-cpu-baseline --no-graph --no-reduced-ab
prof train_r50_b32 --config 3 --steps 10 --warmup 3 --no-cpu-baseline --no-graph --no-overlap --no-stem-ab
S='rocprofv3 --kernel-trace --pmc MfmaUtil GRBM_GUI_ACTIVE | FETCH_SIZE | WRITE_SIZE (separate passes) -- python bench.py'
pmc train_r18_b64 train_b64 "$S --steps 2 --warmup 1 --no-graph; tools/refresh_profiles.sh" --steps 2 --warmup 1 --no-cpu-baseline --no-graph --no-overlap --no-stem-ab
pmc smpl_r18_b65536_fp16x3_lbs smpl_65536 "$S --workload smpl --steps 2 --warmup 1 --no-graph; tools/refresh_profiles.sh" --workload smpl --steps 2 --warmup 1 --no-cpu-baseline --no-graph --no-reduced-ab
pmc smpl_r18_b65536_fp16x3_lbs_p16 smpl_65536_p16 "$S --workload smpl --smpl-precision fp16x3_lbs_p16 --steps 2 --warmup 1 --no-graph; tools/refresh_profiles.sh" --workload smpl --smpl-precision fp16x3_lbs_p16 --steps 2 --warmup 1 --no-cpu-baseline --no-graph
pmc train_r50_b32 train_r50_b32 "$S --config 3 --steps 2 --warmup 1 --no-graph; tools/refresh_profiles.sh" --config 3 --steps 2 --warmup 1 --no-cpu-baseline --no-graph --no-overlap --no-stem-ab
cp $O/pmc_traffic.json $R/profiles/pmc_traffic.json      # bench.py reports roofline.traffic from this file
cd $R
python bench.py 2>/dev/null | tail -1 > $O/${RND}_bench_train_b64.json
python bench.py --conv-precision fp32 --no-cpu-baseline 2>/dev/null | tail -1 > $O/${RND}_bench_train_b64_fp32conv.json
python bench.py --config 1 2>/dev/null | tail -1 > $O/${RND}_bench_fwd_b64.json
python bench.py --config 4 2>/dev/null | tail -1 > $O/${RND}_bench_smpl_1M.json
python bench.py --config 4 --smpl-precision fp32 --no-cpu-baseline 2>/dev/null | tail -1 > $O/${RND}_bench_smpl_1M_fp32.json
python bench.py --config 3 --no-cpu-baseline 2>/dev/null | tail -1 > $O/${RND}_bench_train_r50_b32.json
rm -rf $O/prof_*/ $O/pmc_*/
ls -la $O
