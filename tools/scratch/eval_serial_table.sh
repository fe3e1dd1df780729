#!/bin/bash
# per-kernel table of ONE serial eval forward (8 x 512 x 512, bf16): every kernel alone on the chip
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
D=/tmp/evs_$$
rm -rf $D
EVAL_PROF_STEPS=5 rocprofv3 --kernel-trace -d $D -o p -- python3 $R/tools/eval_prof.py bf16 serial > $D.log 2>&1 || tail -5 $D.log
MS=$(grep "ms/step" $D.log | awk '{print $2}')
DB=$(find $D -name "*results.db" | head -1)
python3 $R/tools/prof_stats.py $DB 5 $MS 60
