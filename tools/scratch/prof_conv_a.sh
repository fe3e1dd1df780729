#!/bin/bash
# device-side duration of the first-layer kernel (host launches of the timing script are slower than the kernel)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pca
rocprofv3 --kernel-trace --stats -d /tmp/pca -o p -- python3 $R/tools/scratch/time_conv_a.py > /tmp/pca.log 2>&1 || tail -5 /tmp/pca.log
DB=$(find /tmp/pca -name "*results.db" | head -1)
python3 $R/tools/rocpd_stats.py $DB | head -6
