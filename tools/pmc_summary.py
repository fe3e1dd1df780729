"""Per-kernel averages of rocprofv3 --pmc counters from counter_collection CSV files.
usage: python tools/pmc_summary.py out.json dir_or_csv [dir_or_csv ...]"""
import csv, glob, json, os, re, sys
out, srcs = sys.argv[1], sys.argv[2:]
agg = {}
for src in srcs:
    files = [src] if src.endswith(".csv") else glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True)
    for f in files:
        for row in csv.DictReader(open(f)):
            name = re.sub(r"\(.*", "", row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", ""))
            k = agg.setdefault(name, {}).setdefault(row["Counter_Name"], [0, 0.0])
            k[0] += 1
            k[1] += float(row["Counter_Value"])
res = {n: {"launches": max(v[0] for v in c.values()), **{cn + "_avg": v[1] / v[0] for cn, v in c.items()}} for n, c in agg.items()}
json.dump(res, open(out, "w"), indent=1, sort_keys=True)
for n, c in sorted(res.items(), key=lambda kv: -sum(v for k, v in kv[1].items() if k.endswith("_avg")))[:12]:
    print(n[:60], {k: round(v, 1) for k, v in c.items()})
