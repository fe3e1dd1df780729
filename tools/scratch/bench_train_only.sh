#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for s in 1 0; do
  for extra in "--no-cqe" ""; do
    MASIC_TRAIN_STREAMS=$s timeout -k 10 400 python bench.py --no-fp8 --no-cpu-baseline --no-codec --no-f32-compare --no-pmc --no-upload --no-trained $extra --train-steps 10 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=d['extras']
print('streams=$s $extra', 'headline', round(d['value']), 'train_step', round(e['train_step']['ms_per_step'],2), 'driver', round(e['train_step_driver_loop']['bf16']['ms_per_step'],2), 'cqe train', e.get('independent_en',{}).get('train_step',{}).get('ms_per_step'))"
  done
done
