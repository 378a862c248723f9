#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for kind in gist deep; do
out=gpurun_out/r3_c5
rm -rf $out && mkdir -p $out
rocprofv3 --kernel-trace --output-format csv -d $out/trace -- python scratch/r3_cfg5_trace.py $kind > $out/log.txt 2>&1
grep MARK $out/log.txt
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r3_c5/trace/*/*_kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
# the last search: from the last init_state_kernel
idx = [i for i, r in enumerate(rows) if "init_state_kernel" in r["Kernel_Name"]][-1]
t0 = int(rows[idx]["Start_Timestamp"])
for r in rows[idx - 6:]:
    n = r["Kernel_Name"].replace("void ", "").replace("amdivf::", "").split("(")[0][:60]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if d > 20: print("%9.1f %9.1f  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, d, n))
PY
rm -rf $out/trace
done
