#!/usr/bin/env python3
"""Per-kernel counters of one scripts/bench_configs.py run under rocprofv3 (profiles/collect_cfg.sh): JSON on stdout.
usage: summarize_cfg.py <dir with trace/ fetch/ write/ sq/ sq2/ misc/ l2/> <cfg> <searches in the run>
Every kernel's figures are sums over its dispatches of the LAST search of the run (its dispatch count / searches).
HBM bytes: FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts a wide (16 B per lane) coalesced stream at half its
bytes (MI355X_MICROARCH.md, HBM): `hbm_read_bytes_x2` is the figure for kernels that stream that way (the filter passes, the
dense tile scan, the matrix product), `hbm_read_bytes_raw` the uncorrected one (gathers of single rows: rescore, coarse_pick)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root, cfg, searches = sys.argv[1], sys.argv[2], int(sys.argv[3])


def short(name):
    return name.split("amdivf::")[1].split("(")[0] if "amdivf::" in name else None


def last_search_rows(ordered):
    """ordered: [(kernel short name, payload)] in dispatch order -> the rows of the LAST search of the run: from the approximate coarse
    ranking's matrix product (coarse_gemm16_kernel / coarse_gemm_kernel: one per search of a large fixed-nprobe call) -- and the few
    small kernels that prepare it -- to the end.  (The same tile kernel also serves k-means while the index is built: counting a
    kernel's dispatches and dividing by the number of searches mixes those in.)"""
    marks = [i for i, (n, _) in enumerate(ordered) if n.startswith("coarse_gemm")]
    if not marks:
        per = {}
        for n, _ in ordered:
            per[n] = per.get(n, 0) + 1
        keep, seen = [], {}
        for n, p in reversed(ordered):
            seen[n] = seen.get(n, 0) + 1
            if seen[n] <= max(1, per[n] // searches):
                keep.append((n, p))
        return list(reversed(keep))
    return ordered[max(0, marks[-1] - 6):]


res = defaultdict(dict)
# kernel trace: durations
f = glob.glob(os.path.join(root, "trace", "**", "*_kernel_trace.csv"), recursive=True)
if f:
    ordered = []
    for r in sorted(csv.DictReader(open(f[0])), key=lambda r: int(r["Dispatch_Id"])):
        n = short(r["Kernel_Name"])
        if n:
            ordered.append((n, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
    by = defaultdict(list)
    for n, ms in last_search_rows(ordered):
        by[n].append(ms)
    for k, ds in by.items():
        res[k]["dispatches_per_search"] = len(ds)
        res[k]["ms_per_search"] = sum(ds)
        res[k]["ms_per_dispatch"] = sum(ds) / len(ds)
for grp in ("fetch", "write", "sq", "sq2", "misc", "l2"):
    f = glob.glob(os.path.join(root, grp, "**", "*_counter_collection.csv"), recursive=True)
    if not f:
        continue
    disp = {}
    for r in csv.DictReader(open(f[0])):
        d = disp.setdefault(int(r["Dispatch_Id"]), {"k": r["Kernel_Name"], "c": defaultdict(float)})
        d["c"][r["Counter_Name"]] += float(r["Counter_Value"])
    ordered = [(short(disp[i]["k"]), disp[i]["c"]) for i in sorted(disp) if short(disp[i]["k"])]
    for k, c in last_search_rows(ordered):
        for name, v in c.items():
            res[k][name] = res[k].get(name, 0.0) + v
out = {"_config": cfg, "_what": "sums over the kernel's dispatches of one search (batch of 10000 queries, nprobe 32)", "_peak_GBps": 8000.0}
step_raw = step_x2 = 0.0
for k, e in sorted(res.items(), key=lambda kv: -kv[1].get("ms_per_search", 0)):
    fe, wr = e.get("FETCH_SIZE", 0.0) * 1024.0, e.get("WRITE_SIZE", 0.0) * 1024.0
    e["hbm_read_bytes_raw"], e["hbm_read_bytes_x2"], e["hbm_write_bytes"] = fe, 2 * fe, wr
    ms = e.get("ms_per_search")
    if ms:
        e["hbm_frac_of_peak_x2"] = (2 * fe + wr) / 1e9 / (ms / 1e3) / 8000.0
        e["hbm_frac_of_peak_raw"] = (fe + wr) / 1e9 / (ms / 1e3) / 8000.0
    wc = e.get("SQ_WAVE_CYCLES")
    if wc:
        e["wave_cycles_active_wait_inst_wait_any"] = [e.get("SQ_ACTIVE_INST_ANY", 0) / wc, e.get("SQ_WAIT_INST_ANY", 0) / wc, e.get("SQ_WAIT_ANY", 0) / wc]
    if e.get("SQ_BUSY_CU_CYCLES"):
        e["mfma_busy_of_cu_busy"] = e.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / e["SQ_BUSY_CU_CYCLES"]
    if e.get("TCC_HIT_sum") is not None and e.get("TCC_HIT_sum", 0) + e.get("TCC_MISS_sum", 0) > 0:
        e["l2_hit_rate"] = e["TCC_HIT_sum"] / (e["TCC_HIT_sum"] + e["TCC_MISS_sum"])
    step_raw += fe + wr
    step_x2 += 2 * fe + wr
    out[k] = e
out["_hbm_bytes_per_search_raw"], out["_hbm_bytes_per_search_x2"] = step_raw, step_x2
print(json.dumps(out, indent=1))
