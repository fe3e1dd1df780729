#!/bin/bash
# usage: bash tools/scratch/prof_tail.sh <script.py> <last N kernels>  -- timeline of the last N kernel dispatches (start offset, duration, name)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
D=/tmp/proft_$$
rm -rf $D
rocprofv3 --kernel-trace -d $D -o p -- python3 $R/$1 > $D.log 2>&1 || tail -5 $D.log
grep "decode ms" $D.log
DB=$(find $D -name "*results.db" | head -1)
python3 - "$DB" "$2" <<'PY'
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
n = int(sys.argv[2])
rows = rows[-n:]
t0 = rows[0][1]
prev_end = t0
for name, s, e in rows:
    name = re.sub(r"\(.*", "", name.replace("(anonymous namespace)::", "").replace("void ", ""))[:60]
    print(f"{(s - t0) / 1e3:9.1f} us  +{(s - prev_end) / 1e3:6.1f} gap  {(e - s) / 1e3:6.1f} us  {name}")
    prev_end = max(prev_end, e)
PY
