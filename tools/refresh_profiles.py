"""Builds the profiles/ artefacts of a round from the files `bash tools/profile_round.sh <tag>` left in gpurun_out/ (see profiles/README.md):
python tools/refresh_profiles.py r02"""
import glob, json, os, re, shutil, sqlite3, statistics as st, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))) + "/"
TAG = sys.argv[1] if len(sys.argv) > 1 else "r02"
sys.path.insert(0, R)
d = json.load(open(R + "gpurun_out/bench_n1.json"))
dom = d["roofline"]["kernel"]
shutil.copy(R + "gpurun_out/bench_n1.json", R + f"profiles/{TAG}_bench_n1.json")
head = subprocess.run(["git", "-C", R, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
# roofline.traffic of the bench line was measured by its own PMC child passes; keep it as the stamped fallback for boxes without rocprofv3
import bench
if d["roofline"].get("traffic"):
    json.dump({"_note": "fallback for bench.py's roofline.traffic when its live rocprofv3 --pmc child passes are unavailable: HBM-side bytes per launch of the "
                        "dominant kernel = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (FETCH_SIZE doubled: 16-byte-per-lane buffer_load ... lds; MI355X_MICROARCH.md, HBM "
                        "section). Only reported while the kernel sources still hash to source_stamp.",
               dom: {"bytes": d["roofline"]["traffic"], "source_stamp": bench.kernel_source_stamp(), "commit": head}},
              open(R + "profiles/pmc_traffic.json", "w"), indent=1)
db_path = glob.glob(R + f"gpurun_out/prof_{TAG}/**/*results.db", recursive=True)[0]
subprocess.run([sys.executable, R + "tools/rocpd_stats.py", db_path, R + f"profiles/{TAG}_kernel_stats.csv"], capture_output=True)
db = sqlite3.connect(db_path)
norm = lambda n: re.sub(r"\(.*", "", n.replace("(anonymous namespace)::", "").replace("void ", ""))
rows = db.execute("select name,start,end from kernels order by start").fetchall()
dd = [(r[2] - r[1]) / 1e3 for r in rows if norm(r[0]) == dom]
json.dump({"symbol": dom, "launches": len(dd), "avg_launch_us_all_launches_of_the_profiled_run": round(st.mean(dd), 1), "median_launch_us": round(st.median(dd), 1),
           "bench_line_avg_launch_us": round(d["roofline"]["avg_launch_ms"] * 1e3, 1), "bench_line_isolated_avg_launch_us": round(d["roofline"]["isolated"]["avg_launch_ms"] * 1e3, 1),
           "note": f"dominant symbol of the bench in the run behind profiles/{TAG}_kernel_stats.csv (graph replays, eager roofline passes and the float32 / fp8 / CQE / training "
                   "extras of that process all launch it or not; the all-launch average of the stats file therefore sits between the bench line's in-situ and isolated figures)"},
          open(R + f"profiles/{TAG}_dominant.json", "w"), indent=1)
# SQ counters (one pass over the eager bf16 forward, bench.py --pmc-child)
sq = glob.glob(R + "gpurun_out/pmc_sq/**/*counter_collection.csv", recursive=True)
if sq:
    subprocess.run([sys.executable, R + "tools/pmc_summary.py", R + f"profiles/{TAG}_sq_counters_raw.json"] + sq, capture_output=True)
    raw = json.load(open(R + f"profiles/{TAG}_sq_counters_raw.json"))
    e = raw.get(dom, {})
    out = {"_note": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE "
                    "-- python3 bench.py --pmc-child (one pass over three eager bf16 forwards, kernel trace off), per-launch averages by tools/pmc_summary.py. Units "
                    "(MI355X_MICROARCH.md): MFMA busy in cycles summed over SIMDs; SQ_WAVE_CYCLES / SQ_WAIT_* in quad-cycles summed over waves; LDS counters in cycles summed over CUs.",
           "commit": head, "dominant": dom, "dominant_counters": e, "kernels": raw}
    if e.get("SQ_WAVE_CYCLES_avg"):
        w = e["SQ_WAVE_CYCLES_avg"]
        out["dominant_reading"] = {"waves_parked_frac (SQ_WAIT_ANY / SQ_WAVE_CYCLES)": round(e.get("SQ_WAIT_ANY_avg", 0) / w, 3),
                                   "issue_stalled_frac (SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES)": round(e.get("SQ_WAIT_INST_ANY_avg", 0) / w, 3),
                                   "lds_issue_stall_frac": round(e.get("SQ_WAIT_INST_LDS_avg", 0) / w, 3),
                                   "mfma_busy_over_busy_cycles (x4 SIMDs normalisation left to the reader)": round(e.get("SQ_VALU_MFMA_BUSY_CYCLES_avg", 0) / max(e.get("SQ_BUSY_CYCLES_avg", 1), 1), 3),
                                   "lds_bank_conflict_over_lds_active": round(e.get("SQ_LDS_BANK_CONFLICT_avg", 0) / max(e.get("SQ_LDS_IDX_ACTIVE_avg", 1), 1), 4)}
    json.dump(out, open(R + f"profiles/{TAG}_sq_counters.json", "w"), indent=1, sort_keys=True)
    os.remove(R + f"profiles/{TAG}_sq_counters_raw.json")
for name, log, n in (("train", "prof_train", 3), ("cqe", "prof_cqe", 3), ("cqe_train", "prof_cqetrain", 3)):
    dbs = glob.glob(R + f"gpurun_out/{log}/**/*results.db", recursive=True)
    m = re.search(r"ms/step ([0-9.]+)", open(R + f"gpurun_out/{log}.log").read()) if os.path.exists(R + f"gpurun_out/{log}.log") else None
    if dbs and m:
        txt = subprocess.run([sys.executable, R + "tools/prof_stats.py", dbs[0], str(n), m.group(1), "60"], capture_output=True, text=True).stdout
        open(R + f"profiles/{TAG}_{name}_step_kernels.txt", "w").write(f"# rocprofv3 --kernel-trace -- python3 tools/{'train_prof.py bf16' if name == 'train' else ('cqe_prof.py bf16 train' if name == 'cqe_train' else 'cqe_prof.py bf16')} (commit {head}); ms/step under the profiler {m.group(1)}" + (" -- the step runs on three streams (MASIC.py: _forward_graph): kernels share the chip, so a kernel's duration here is longer than alone and the column adds up to more than the step; MASIC_TRAIN_STREAMS=0 gives the one-stream table" if name == "train" else "") + "\n" + txt)
print(dom, "value", d["value"], "frac", d["roofline"]["frac"], "traffic", d["roofline"].get("traffic"))
