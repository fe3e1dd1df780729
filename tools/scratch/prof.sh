#!/bin/bash
# usage: bash tools/scratch/prof.sh <script.py> [grep pattern]  -- device-side kernel durations of a script (rocprofv3 kernel trace)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
D=/tmp/prof_$$
rm -rf $D
rocprofv3 --kernel-trace --stats -d $D -o p -- python3 $R/$1 > $D.log 2>&1 || tail -5 $D.log
grep -v "amdgpu.ids\|rocprofv3\|^W2026\|^E2026" $D.log | tail -${3:-12}
DB=$(find $D -name "*results.db" | head -1)
python3 $R/tools/rocpd_stats.py $DB | grep -i "${2:-.}" | head -8
