"""Per-kernel ms/step of the last `nsteps` identical steps in a rocprofv3 rocpd database: python tools/prof_stats.py results.db nsteps [top]"""
import sqlite3, re, sys, collections
db = sqlite3.connect(sys.argv[1]); nsteps = int(sys.argv[2]); top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
rows = db.execute("select name,start,end from kernels order by start").fetchall()
norm = lambda n: re.sub(r"\(.*", "", n.replace("(anonymous namespace)::", "").replace("void ", ""))
names = [norm(r[0]) for r in rows]
# period = distance between the last two occurrences of the rarest kernel that occurs >= nsteps + 1 times
c = collections.Counter(names)
cand = [k for k, v in c.items() if v >= nsteps + 1]
key = min(cand, key=lambda k: c[k])
idx = [i for i, n in enumerate(names) if n == key]
per = c[key] // (nsteps + 2) if c[key] % (nsteps + 2) == 0 else 1
period = idx[-1] - idx[-1 - per]
sel = rows[len(rows) - nsteps * period:]
agg = collections.defaultdict(lambda: [0, 0])
for nm, s, e in sel:
    a = agg[norm(nm)]; a[0] += 1; a[1] += e - s
tot = sum(v[1] for v in agg.values())
print(f"kernels/step {period}  kernel ms/step {tot / nsteps / 1e6:.3f}  span ms/step {(sel[-1][2] - sel[0][1]) / nsteps / 1e6:.3f}")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"{v[1] / nsteps / 1e6:8.3f} ms/step {v[0] / nsteps:7.1f} calls  {k[:120]}")
