"""Per-kernel ms/step over the last `nsteps` steps of a rocprofv3 rocpd database, the steps being the last nsteps * ms_per_step
milliseconds of the trace:   python tools/prof_stats.py results.db nsteps ms_per_step [top]"""
import sqlite3, re, sys, collections
db = sqlite3.connect(sys.argv[1]); nsteps = int(sys.argv[2]); ms = float(sys.argv[3]); top = int(sys.argv[4]) if len(sys.argv) > 4 else 40
rows = db.execute("select name,start,end from kernels order by start").fetchall()
norm = lambda n: re.sub(r"\(.*", "", n.replace("(anonymous namespace)::", "").replace("void ", ""))
t_end = max(r[2] for r in rows)
cut = t_end - nsteps * ms * 1e6
sel = [r for r in rows if r[1] >= cut]
agg = collections.defaultdict(lambda: [0, 0])
for nm, s, e in sel:
    a = agg[norm(nm)]; a[0] += 1; a[1] += e - s
tot = sum(v[1] for v in agg.values())
print(f"kernels/step {len(sel) / nsteps:.1f}  kernel ms/step {tot / nsteps / 1e6:.3f}  window ms/step {ms:.3f}")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"{v[1] / nsteps / 1e6:8.4f} ms/step {v[0] / nsteps:6.1f} calls {v[1] / v[0] / 1e3:8.1f} us avg  {k[:110]}")
