#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" "FETCH_SIZE" ; do
out=gpurun_out/r3_c5pmc
rm -rf $out && mkdir -p $out
timeout 600 rocprofv3 --pmc $grp --output-format csv -d $out/p -- python3 scratch/r3_cfg5_trace.py gist > $out/log.txt 2>&1
grep MARK $out/log.txt | head -3
python - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/r3_c5pmc/p/*/*_counter_collection.csv")
if not f: print("no counters", open("gpurun_out/r3_c5pmc/log.txt").read()[-600:]); raise SystemExit
rows = list(csv.DictReader(open(f[0])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in rows:
    n = r["Kernel_Name"]
    if "scan_filter" in n or "scan_tiles_kernel<1, 1" in n or "rescore" in n:
        k = n.replace("void ","").replace("amdivf::","").split("(")[0]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
        if r["Counter_Name"] == rows[0]["Counter_Name"]: cnt[k] += 1
for k, v in agg.items():
    print(k, "dispatches", cnt[k], {c: "%.4g" % (x / max(cnt[k],1)) for c, x in v.items()})
PY
rm -rf $out/p
done
