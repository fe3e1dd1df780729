"""Every kernel launch of the LAST step of a rocprofv3 rocpd database, in start order:   python tools/prof_timeline.py results.db ms_per_step
(the last step = the last ms_per_step milliseconds of the trace)"""
import sqlite3, re, sys
db = sqlite3.connect(sys.argv[1]); ms = float(sys.argv[2])
rows = db.execute("select name,start,end,grid_x,grid_y,grid_z,workgroup_x from kernels order by start").fetchall()
norm = lambda n: re.sub(r"\(.*", "", n.replace("(anonymous namespace)::", "").replace("void ", ""))
t_end = max(r[2] for r in rows)
sel = [r for r in rows if r[1] >= t_end - ms * 1e6]
t0 = sel[0][1]
print(f"{len(sel)} launches, span {(t_end - t0) / 1e3:.1f} us, kernel time {sum(r[2] - r[1] for r in sel) / 1e3:.1f} us")
for nm, s, e, gx, gy, gz, wx in sel:
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} us  wgs {gx // max(wx, 1) * gy * gz:6d}  {norm(nm)[:100]}")
