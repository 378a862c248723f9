#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
export AUNCEL_AMD_NO_BYTES=1
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"; do
OUT=/tmp/fpmc; rm -rf $OUT; mkdir -p $OUT
timeout 600 rocprofv3 --pmc $grp --output-format csv -d $OUT -- python3 bench.py --no-cpu --no-legs --in-flight 1 --steps 4 --warmup 2 > $OUT/run.log 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/*/*_counter_collection.csv")
rows = list(csv.DictReader(open(f[0])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    n = r["Kernel_Name"]
    if "scan_filter" in n or "rescore" in n or "scan_tiles_kernel<1, 1, 1>" in n:
        agg[n.replace("void ","").replace("amdivf::","").split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    for c, xs in v.items():
        xs = xs[-12:]
        print(k, c, "last dispatches:", ["%.3g" % x for x in xs[-6:]])
PY
done
