#!/bin/bash
# A/B of the training step and the forward with the encoder convolutions on the exact-fp32 chain vs the bf16x3 route
mkdir -p gpurun_out/x3
for p in fp32 bf16x3; do
  python bench.py --no-cpu-baseline --no-stem-ab --conv-precision $p > gpurun_out/x3/train_$p.json 2> gpurun_out/x3/train_$p.err
  python bench.py --workload fwd --no-cpu-baseline --conv-precision $p > gpurun_out/x3/fwd_$p.json 2> gpurun_out/x3/fwd_$p.err
done
python - <<'PY'
import json
for w in ('train','fwd'):
    for p in ('fp32','bf16x3'):
        try:
            d=json.loads(open('gpurun_out/x3/%s_%s.json'%(w,p)).read().strip().splitlines()[-1])
            print(w,p,d['value'],d['ms_per_step'],d.get('final_loss'),d['roofline'])
        except Exception as e:
            print(w,p,'FAILED',e); print(open('gpurun_out/x3/%s_%s.err'%(w,p)).read()[-1500:])
PY
