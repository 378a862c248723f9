#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for c in 5 3; do
out=gpurun_out/r3_cfgtrace$c
rm -rf $out && mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python scripts/bench_configs.py --cfg $c --sample 8 --ref-sample 0 > $out/log.txt 2>&1
python - <<PY
import csv, glob
f = glob.glob("$out/trace/*/*_kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
print("== cfg $c")
for r in rows[:14]:
    n = r["Name"].replace("void ", "").replace("amdivf::", "")[:90]
    print("%10.1f us total %6s calls %9.1f avg  %s" % (float(r["TotalDurationNs"]) / 1e3, r["Calls"], float(r["AverageNs"]) / 1e3, n))
PY
rm -rf $out/trace
done
