#!/bin/bash
# kernel trace of the one-batch bench: per-kernel average durations
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r3_trace
rm -rf $out && mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python bench.py --no-cpu --no-legs --in-flight 1 --steps 5 > $out/trace.log 2>&1
python - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/r3_trace/trace/*/*_kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
# last 5 steps: take the last 5 occurrences of sort_prefix/sort_rows as step marks
names = [r["Kernel_Name"] for r in rows]
marks = [i for i, n in enumerate(names) if "sort_prefix_kernel" in n or "sort_rows_kernel" in n]
start = marks[-5] - 3
agg = collections.OrderedDict()
for r in rows[start:]:
    n = r["Kernel_Name"].split("amdivf::")[-1].split("(")[0][:70]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    a = agg.setdefault(n, [0, 0.0])
    a[0] += 1; a[1] += d
tot = sum(v[1] for v in agg.values())
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{t/5:9.1f} us/step  {c/5:5.1f} calls/step  {t/c:8.1f} us/call  {n}")
print("sum of kernels per step: %.1f us" % (tot / 5))
# timeline of the last step
last = marks[-1] - 3
t0 = int(rows[last]["Start_Timestamp"])
for r in rows[last:]:
    n = r["Kernel_Name"].split("amdivf::")[-1].split("(")[0][:60]
    print("%8.1f %8.1f  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, n))
PY
rm -rf $out/trace
