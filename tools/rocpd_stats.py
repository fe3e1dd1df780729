"""Per-kernel summary (calls, total, avg, min, max, %) from a rocprofv3 rocpd SQLite file, as the CSV rocprofv3 --stats prints.
usage: python tools/rocpd_stats.py results.db [out.csv] [--last-frac F]   (F: only dispatches in the last F of the run)"""
import sqlite3, sys, re
args = [a for a in sys.argv[1:] if not a.startswith("--")]
frac = float(sys.argv[sys.argv.index("--last-frac") + 1]) if "--last-frac" in sys.argv else 1.0
if "--last-frac" in sys.argv:
    args = [a for a in args if a != sys.argv[sys.argv.index("--last-frac") + 1]]
db = sqlite3.connect(args[0])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
t0, t1 = rows[0][1], rows[-1][2]
cut = t1 - (t1 - t0) * frac
agg = {}
for name, s, e in rows:
    if s < cut:
        continue
    name = re.sub(r"\(.*", "", name.replace("(anonymous namespace)::", "").replace("void ", ""))
    a = agg.setdefault(name, [0, 0, 10**18, 0])
    d = e - s
    a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
tot = sum(a[1] for a in agg.values())
lines = ["Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs"]
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    lines.append(f'"{k}",{a[0]},{a[1]},{a[1] / a[0]:.1f},{100.0 * a[1] / tot:.2f},{a[2]},{a[3]}')
out = "\n".join(lines)
if len(args) > 1:
    open(args[1], "w").write(out + "\n")
print(out)
print(f"# total kernel ns {tot}, span ns {t1 - max(cut, t0):.0f}", file=sys.stderr)
