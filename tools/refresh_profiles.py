"""Builds the profiles/ artefacts from the files a GPU run left in gpurun_out/ (bench_n1.json, prof_r01/r01_results.db,
pmc_fetch/, pmc_write/): see profiles/README.md."""
import json, os, re, shutil, sqlite3, statistics as st, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))) + "/"
d = json.load(open(R + "gpurun_out/bench_n1.json"))
dom = d["roofline"]["kernel"]
subprocess.run([sys.executable, R + "tools/pmc_summary.py", R + "profiles/r01_pmc_summary.json", R + "gpurun_out/pmc_fetch", R + "gpurun_out/pmc_write"], capture_output=True)
pm = json.load(open(R + "profiles/r01_pmc_summary.json"))
e = pm[dom]
traffic = (2 * e["FETCH_SIZE_avg"] + e["WRITE_SIZE_avg"]) * 1024
json.dump({"_note": "HBM-side bytes per launch of the bench's dominant kernel = (2 x FETCH_SIZE + WRITE_SIZE) x 1024, averaged over its launches in separate `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes of `python bench.py --steps 2 --warmup 1 --no-cpu-baseline --train-steps 0 --no-f32-compare --no-graph` (profiles/r01_pmc_summary.json). FETCH_SIZE is doubled because every load of this kernel is a 16-byte-per-lane buffer_load ... lds (MI355X_MICROARCH.md, HBM section: the counter reports 1/2 for wide coalesced reads); WRITE_SIZE at face value (16-byte-per-lane stores).", dom: traffic},
          open(R + "profiles/pmc_traffic.json", "w"), indent=1)
shutil.copy(R + "gpurun_out/bench_n1.json", R + "profiles/r01_bench_n1.json")
subprocess.run([sys.executable, R + "tools/rocpd_stats.py", R + "gpurun_out/prof_r01/r01_results.db", R + "profiles/r01_kernel_stats.csv"], capture_output=True)
db = sqlite3.connect(R + "gpurun_out/prof_r01/r01_results.db")
rows = db.execute("select name,start,end from kernels order by start").fetchall()
dd = [(r[2] - r[1]) / 1e3 for r in rows if re.sub(r"\(.*", "", r[0].replace("(anonymous namespace)::", "").replace("void ", "")) == dom]
n = int(round(d["roofline"]["launches_per_step"]))
fw = [dd[i:i + n] for i in range(0, len(dd), n)]
steps, warm = d["steps"], d["warmup"]
sec = {"graph_capture_warmup_eager": fw[0:2], "graph_replays": fw[2:2 + warm + steps], "eager_roofline_pass": fw[2 + warm + steps:2 + warm + steps + 1 + steps]}
out = {k: {"forwards": len(v), "avg_launch_us": round(st.mean([x for f in v for x in f]), 1)} for k, v in sec.items() if v}
json.dump({"symbol": dom, "launches_per_forward": n, "sections": out,
           "note": "per-section average launch duration of the dominant symbol in the run behind profiles/r01_kernel_stats.csv (order of its launches: 2 eager warm-up forwards of the graph capture, warm-up + timed graph replays, 1 + steps eager forwards of the roofline pass; the symbol-stream / float32 comparison and the training steps launch it a few more times or not at all). Inside graph replays the analysis transforms of the two views run concurrently on two streams and share the CUs, so the same launches take longer there; bench.py's roofline.avg_launch_ms is the eager pass."},
          open(R + "profiles/r01_dominant_sections.json", "w"), indent=1)
print(dom, "traffic MB", traffic / 1e6, out, "bench avg_launch_ms", d["roofline"]["avg_launch_ms"], "value", d["value"])
