#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
D=/tmp/proftr_$$
rm -rf $D
TRAIN_PROF_STEPS=4 rocprofv3 --kernel-trace -d $D -o p -- python3 $R/tools/train_prof.py bf16 > $D.log 2>&1 || tail -5 $D.log
grep "ms/step" $D.log
DB=$(find $D -name "*results.db" | head -1)
python3 - "$DB" <<'PY'
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
# last 4 steps: find by total count / 6 steps (2 warm + 4 timed) -> take the last 4/6 of rows after the warm-up boundary by time
t_end = rows[-1][2]
# step boundaries unknown: use the last 60 % of the dispatches' time span as 4 steps is fragile; instead count kernels named FusedAdam (2 per step)
adam = [i for i, r in enumerate(rows) if "FusedOptimizer" in r[0] or "multi_tensor_apply" in r[0]]
per = {}
first = None
# steps end at every second Adam group; take rows after the 4th-last main Adam
starts = [i for i in adam]
n = len(rows)
# simple: take the last 4/6 of kernel count
cut = n - (n * 4) // 6
sel = rows[cut:]
agg = {}
for name, s, e in sel:
    name = re.sub(r"\(.*", "", name.replace("(anonymous namespace)::", "").replace("void ", ""))[:70]
    a = agg.setdefault(name, [0, 0])
    a[0] += 1; a[1] += e - s
tot = sum(a[1] for a in agg.values())
print(f"kernels/step {len(sel) / 4:.0f}  kernel ms/step {tot / 4e6:.3f}  window ms/step {(sel[-1][2] - sel[0][1]) / 4e6:.3f}")
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f"{a[1] / 4e6:8.4f} ms/step {a[0] / 4:6.1f} calls {a[1] / a[0] / 1e3:8.1f} us avg  {k}")
PY
