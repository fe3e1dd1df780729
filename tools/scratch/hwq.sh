#!/bin/bash
# does the number of hardware queues HIP multiplexes its streams onto matter for the multi-stream graph / training step?
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for q in default 2 4 8; do
  if [ $q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
  timeout -k 10 400 python bench.py --no-cpu-baseline --no-codec --no-f32-compare --no-pmc --no-upload --no-trained --no-cqe --train-steps 10 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=d['extras']
print('GPU_MAX_HW_QUEUES=$q', 'headline', round(d['value']), 'fp8', round(e['fp8_path']['value']), 'no_lookahead', round(e['no_lookahead']['value']), 'train_step', round(e['train_step']['ms_per_step'],2), 'driver', round(e['train_step_driver_loop']['bf16']['ms_per_step'],2))"
done
