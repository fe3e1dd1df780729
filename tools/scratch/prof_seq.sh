#!/bin/bash
# usage: bash tools/scratch/prof_seq.sh <script.py> <kernel substring>  -- per-dispatch durations (us) of one kernel, in dispatch order
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
D=/tmp/profs_$$
rm -rf $D
rocprofv3 --kernel-trace -d $D -o p -- python3 $R/$1 > $D.log 2>&1 || tail -5 $D.log
DB=$(find $D -name "*results.db" | head -1)
python3 - "$DB" "$2" <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
d = [(e - s) / 1e3 for n, s, e in rows if sys.argv[2] in n]
print(len(d), "dispatches:", " ".join(f"{v:.1f}" for v in d))
PY
