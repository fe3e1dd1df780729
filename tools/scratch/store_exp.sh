#!/bin/bash
# Transposed 128 -> 128 + IGDN layer (conv_f16k<2,2,2,5,1,true,4,2>): what would whole-line stores be worth?  
# MASIC_F16K_STORE_EXP = 0 shipped stores, 1 stores skipped at run time, 2 phase-planar positions (whole contiguous records per wave).
R=${GRAFT_REPO_ROOT:-/root/repo}
for e in 0 2 1; do
  export MASIC_F16K_STORE_EXP=$e
  echo "== MASIC_F16K_STORE_EXP=$e: serial eval forward, per-dispatch us of the transposed GDN kernel, then of the strided GDN kernel"
  bash $R/tools/scratch/prof_seq.sh "tools/eval_prof.py bf16 serial" "conv_f16k<2, 2, 2, 5, 1, true, 4, 2" || exit 1
  echo "   graph replay (no profiler):"; python3 $R/tools/eval_prof.py bf16 | tail -1
done
