"""per-kernel averages of a rocprofv3 --pmc run (one row per dispatch and counter in *_counter_collection.csv)
usage: python profiles/pmc_by_kernel.py <dir> [last_n_dispatches_per_kernel]"""
import csv, glob, sys
from collections import defaultdict
n_last = int(sys.argv[2]) if len(sys.argv) > 2 else 5
f = sorted(glob.glob(sys.argv[1] + "/**/*_counter_collection.csv", recursive=True))[-1]
disp = {}
for r in csv.DictReader(open(f)):
    d = disp.setdefault(int(r["Dispatch_Id"]), {"k": r["Kernel_Name"], "c": defaultdict(float), "t": int(r["End_Timestamp"]) - int(r["Start_Timestamp"])})
    d["c"][r["Counter_Name"]] += float(r["Counter_Value"])
by = defaultdict(list)
for k in sorted(disp):
    n = disp[k]["k"]
    if "amdivf::" not in n:
        continue
    by[n.split("amdivf::")[1].split("(")[0]].append(disp[k])
for n, ds in sorted(by.items(), key=lambda kv: -sum(d["t"] for d in kv[1][-n_last:])):
    ds = ds[-n_last:]
    tot = defaultdict(float)
    for d in ds:
        for c, v in d["c"].items():
            tot[c] += v
    print(f"{n[:60]:60s} n={len(ds)} avg_us={sum(d['t'] for d in ds)/len(ds)/1e3:9.1f} " + " ".join(f"{c}={v/len(ds):.4g}" for c, v in sorted(tot.items())))
