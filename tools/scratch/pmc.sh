#!/bin/bash
# usage: bash tools/scratch/pmc.sh <script.py> <kernel substring>  -- FETCH_SIZE / WRITE_SIZE (KiB per launch, averaged) of one kernel, two passes
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  D=/tmp/pmc_$$_$C
  rm -rf $D
  rocprofv3 --pmc $C --output-format csv -d $D -o p -- python3 $R/$1 > $D.log 2>&1 || tail -3 $D.log
  F=$(find $D -name "*counter_collection.csv" | head -1)
  python3 - "$F" "$2" "$C" <<'PY'
import csv, sys
vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r["Kernel_Name"] and r["Counter_Name"] == sys.argv[3]]
print(sys.argv[3], len(vals), "launches (KiB each, dispatch order):", " ".join(f"{v:.0f}" for v in vals[:12]), "...", " ".join(f"{v:.0f}" for v in vals[-70:-60]))
PY
done
