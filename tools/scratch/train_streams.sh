#!/bin/bash
# training step with the right view's front part on a side stream (MASIC_TRAIN_STREAMS=1) and the parameter gradients on another
# (MASIC_WGRAD_STREAM=1): eager, against the one-stream step
R=${GRAFT_REPO_ROOT:-/root/repo}
for cfg in "0 0" "1 0" "0 1" "1 1"; do
  set -- $cfg
  echo "== MASIC_TRAIN_STREAMS=$1 MASIC_WGRAD_STREAM=$2"
  MASIC_TRAIN_STREAMS=$1 MASIC_WGRAD_STREAM=$2 TRAIN_PROF_STEPS=10 timeout -k 10 200 python3 $R/tools/train_prof.py bf16 2>&1 | tail -1 || exit 1
done
