"""GPU idle time inside the last `nsteps` steps of a rocprofv3 rocpd database (all streams merged): busy = union of the kernel
intervals; gaps are listed by size class and the largest ones with the kernels on either side.
    python tools/prof_gaps.py results.db nsteps ms_per_step"""
import sqlite3, re, sys
db = sqlite3.connect(sys.argv[1]); nsteps = int(sys.argv[2]); ms = float(sys.argv[3])
rows = db.execute("select name,start,end from kernels order by start").fetchall()
norm = lambda n: re.sub(r"\(.*", "", n.replace("(anonymous namespace)::", "").replace("void ", ""))[:60]
t_end = max(r[2] for r in rows)
sel = [r for r in rows if r[1] >= t_end - nsteps * ms * 1e6]
busy, gaps, cur_end, prev = 0, [], sel[0][1], sel[0]
for r in sel:
    if r[1] > cur_end:
        gaps.append((r[1] - cur_end, norm(prev[0]), norm(r[0])))
        busy += 0
    if r[2] > cur_end:
        busy += r[2] - max(r[1], cur_end)
        cur_end, prev = r[2], r
span = cur_end - sel[0][1]
print(f"{len(sel) / nsteps:.0f} launches/step, span {span / nsteps / 1e6:.3f} ms/step, busy {busy / nsteps / 1e6:.3f} ms/step, idle {(span - busy) / nsteps / 1e6:.3f} ms/step")
for lo, hi in ((0, 2e3), (2e3, 5e3), (5e3, 20e3), (20e3, 100e3), (100e3, 1e12)):
    g = [x[0] for x in gaps if lo <= x[0] < hi]
    print(f"  gaps {lo / 1e3:5.0f} .. {hi / 1e3:7.0f} us: {len(g) / nsteps:7.1f} per step, {sum(g) / nsteps / 1e6:7.3f} ms/step")
for g in sorted(gaps, reverse=True)[:12]:
    print(f"  {g[0] / 1e3:8.1f} us after {g[1]} before {g[2]}")
