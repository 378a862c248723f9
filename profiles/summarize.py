#!/usr/bin/env python3
"""Summarise the rocprofv3 output of profiles/collect.sh for bench.py.

usage: summarize.py <gpurun_out/prof_TAG> <TAG> [outdir]   -> writes <outdir or profiles>/<TAG>_summary.md and <TAG>_pmc.json

The timed region of bench.py is its last `steps` steps.  Every step starts with the coarse quantisation, whose
row sort (sort_rows_kernel) is dispatched exactly once per step, so the dispatches from the steps-th-from-last
sort_rows_kernel on are the timed steps (warm-up, trace training and the hyper-parameter search come before).
List scans are the scan_tiles_kernel dispatches other than each step's first one (the coarse distances use the
same kernel).  HBM bytes: FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts a wide (16 B/lane)
coalesced stream at half its bytes (MI355X_MICROARCH.md, HBM), so the scan's read side is doubled."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root, tag = sys.argv[1], sys.argv[2]
here = sys.argv[3] if len(sys.argv) > 3 else os.path.dirname(os.path.abspath(__file__))  # where the summaries go
os.makedirs(here, exist_ok=True)


def bench_json(log):
    j = None
    for line in open(log, errors="replace"):
        i = line.find('{"metric"')
        if i >= 0:
            j = json.loads(line[i:])
    return j


def short(name):
    return name.split("amdivf::")[1].split("(")[0] if "amdivf::" in name else None


def is_sort(n):
    return n.startswith("sort_rows_kernel") or n.startswith("sort_prefix_kernel")


def timed_sequences(rows, j, key_name):
    """rows: dicts in dispatch order -> dispatch sequences of the timed steps, one per host thread.
    --in-flight 1: the main thread; the timed steps start at the steps-th-from-last coarse ranking.
    --in-flight N: the timed steps are issued by N worker threads created for the timed region (the warm-up has its
    own), i.e. the N thread ids that appear last; everything they dispatch belongs to timed steps."""
    steps, nfl = j["steps"], j["config"].get("in_flight", 1)
    eng = [r for r in rows if short(r[key_name])]
    if nfl > 1:
        first = {}
        for i, r in enumerate(eng):
            first.setdefault(r["Thread_Id"], i)
        workers = sorted(first, key=first.get)[-nfl:]
        return [[r for r in eng if r["Thread_Id"] == t] for t in workers]
    marks = [i for i, r in enumerate(eng) if is_sort(short(r[key_name]))]
    # a step's coarse scan precedes its ranking dispatch by one engine dispatch (pack + scan): back up to the pack
    start = marks[-steps]
    while start > 0 and not short(eng[start][key_name]).startswith("pack_queries_kernel"):
        start -= 1
    return [eng[start:]]


def is_list_scan(seq, i, key_name):
    """scan_tiles dispatch that is not the coarse one (the coarse scan is the one right before sort_rows)"""
    n = short(seq[i][key_name])
    if not n.startswith("scan_tiles_kernel"):
        return False
    for j in range(i + 1, len(seq)):
        m = short(seq[j][key_name])
        if m.startswith("scan_tiles_kernel"):
            continue
        return not is_sort(m)
    return True


lines = []
tj = bench_json(os.path.join(root, "trace.log"))
steps = tj["steps"]
tr = sorted(csv.DictReader(open(glob.glob(os.path.join(root, "trace", "*", "*_kernel_trace.csv"))[0])),
            key=lambda r: int(r["Dispatch_Id"]))
agg = defaultdict(list)
for seq in timed_sequences(tr, tj, "Kernel_Name"):
    for i, r in enumerate(seq):
        n = short(r["Kernel_Name"])
        if n.startswith("scan_tiles_kernel"):
            n += " [lists]" if is_list_scan(seq, i, "Kernel_Name") else " [coarse]"
        agg[n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
lines.append(f"# {tag}: bench.py under rocprofv3 (MI355X), timed region = last {steps} steps\n")
lines.append("## kernel trace (`rocprofv3 --kernel-trace --stats`)\n")
lines.append("| kernel | calls | calls/step | total ms | ms/step | avg ms | max ms |")
lines.append("|---|---|---|---|---|---|---|")
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    lines.append(f"| {k} | {len(v)} | {len(v)/steps:.1f} | {sum(v):.3f} | {sum(v)/steps:.3f} | {sum(v)/len(v):.4f} | {max(v):.4f} |")
def is_scan(k):
    return k.endswith("[lists]") or k.startswith("scan_mfma")


scan_ms = sum(sum(v) for k, v in agg.items() if is_scan(k))
rf = tj["roofline"]
nl = rf["launches_per_step"] * steps
lines.append("")
lines.append(f"list-scan launches (one per round, up to four tile shapes each, side by side on four streams): "
             f"{nl:.0f}; summed kernel durations {scan_ms:.2f} ms = {scan_ms/nl:.4f} ms per launch if run back to back; "
             f"bench.py's HIP events around each launch: avg_launch_ms = {rf['avg_launch_ms']:.4f} "
             f"(shapes overlap, so the event span is <= the sum)")
lines.append(f"bench line of this run: value {tj['value']:.0f} q/s, {tj['ms_per_step']:.2f} ms/step, achieved "
             f"{rf['achieved']:.0f} GB/s algorithmic, recall@10 {tj['config']['recall_at_10_mean']:.4f}, "
             f"nprobe mean {tj['config']['nprobe_mean']:.1f}")

pmc = {}
for grp in ("fetch", "write", "sq", "misc", "l2", "l1", "sq2"):
    f = glob.glob(os.path.join(root, grp, "*", "*_counter_collection.csv"))
    if not f:
        continue
    j = bench_json(os.path.join(root, grp + ".log"))
    rows = list(csv.DictReader(open(f[0])))
    # one row per (dispatch, counter): rebuild dispatch order
    disp = {}
    for r in rows:
        d = disp.setdefault(int(r["Dispatch_Id"]), {"Kernel_Name": r["Kernel_Name"], "Thread_Id": r["Thread_Id"], "c": {},
                                                    "t": int(r["End_Timestamp"]) - int(r["Start_Timestamp"])})
        d["c"][r["Counter_Name"]] = d["c"].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    order = [disp[k] for k in sorted(disp)]
    for seq in timed_sequences(order, j, "Kernel_Name"):
        for i, r in enumerate(seq):
            n = short(r["Kernel_Name"])
            if n.startswith("scan_tiles_kernel"):
                n = "scan_tiles_kernel [lists]" if is_list_scan(seq, i, "Kernel_Name") else "scan_tiles_kernel [coarse]"
            e = pmc.setdefault(n, defaultdict(float))  # (every kernel by its full name: dense and threshold scans apart)
            for c, v in r["c"].items():
                e[c] += v
            e["_dispatches_" + grp] += 1
            e["_ns_" + grp] += r["t"]
            e["_steps_" + grp] = j["steps"]
            if is_scan(n):
                e["_alg_bytes_per_launch"] = j["roofline"]["algorithmic_bytes_per_launch"]
lines.append("\n## PMC (separate passes: FETCH_SIZE | WRITE_SIZE | SQ_* | GRBM/LDS)\n")
lines.append("| kernel | FETCH_SIZE KiB | WRITE_SIZE KiB | HBM GB/step (reads x2 for the scan) | VALU insts | wave cycles: active / wait_inst / wait_any | LDS bank conflicts | clock GHz |")
lines.append("|---|---|---|---|---|---|---|---|")
out = {}
step_bytes = 0.0


def wide_reader(k):
    """kernels whose reads are 16-byte-per-lane coalesced streams (FETCH_SIZE counts those at half their bytes on gfx950,
    MI355X_MICROARCH.md HBM): the scans, the fp32 filter, the selection of a dense round (rows of distances)"""
    return k.startswith(("scan_tiles_kernel", "scan_lanes_kernel", "scan_mfma", "scan_filter")) or k.startswith("select_sorted_kernel<true, false") \
        or k.startswith("select_sorted_kernel<false, false")


for k, e in sorted(pmc.items(), key=lambda kv: -kv[1].get("FETCH_SIZE", 0)):
    st = e.get("_steps_fetch", steps)
    corr = 2.0 if wide_reader(k) else 1.0
    hbm = (e.get("FETCH_SIZE", 0) * corr + e.get("WRITE_SIZE", 0)) * 1024
    wc = e.get("SQ_WAVE_CYCLES", 0)
    frac = (f"{e.get('SQ_ACTIVE_INST_ANY', 0)/wc:.2f} / {e.get('SQ_WAIT_INST_ANY', 0)/wc:.2f} / {e.get('SQ_WAIT_ANY', 0)/wc:.2f}" if wc else "-")
    clk = e.get("GRBM_GUI_ACTIVE", 0) / 8 / e["_ns_misc"] if e.get("_ns_misc") else 0
    lines.append(f"| {k} | {e.get('FETCH_SIZE', 0):.4g} | {e.get('WRITE_SIZE', 0):.4g} | {hbm/st/1e9:.3f} | {e.get('SQ_INSTS_VALU', 0):.4g} | {frac} | "
                 f"{e.get('SQ_LDS_BANK_CONFLICT', 0):.3g} | {clk:.2f} |")
    nd = max(e.get("_dispatches_fetch", 0), 1)
    step_bytes += hbm / st
    out[k] = {"fetch_size_kib": e.get("FETCH_SIZE", 0), "write_size_kib": e.get("WRITE_SIZE", 0), "read_correction": corr,
              "hbm_bytes_per_step": hbm / st, "steps": st, "dispatches": nd, "hbm_bytes_per_dispatch": hbm / nd,
              "avg_ms_fetch_pass": e.get("_ns_fetch", 0) / nd / 1e6}
    if is_scan(k):
        out[k]["algorithmic_bytes_per_launch"] = e.get("_alg_bytes_per_launch", 0)
lines.append("")
for k, o in out.items():
    if is_scan(k) and o["avg_ms_fetch_pass"] > 0:
        gbps = o["hbm_bytes_per_dispatch"] / 1e9 / (o["avg_ms_fetch_pass"] / 1e3)
        lines.append(f"{k}: {o['hbm_bytes_per_dispatch']/1e9:.3f} GB per dispatch (FETCH x {o['read_correction']:.0f} + WRITE) in "
                     f"{o['avg_ms_fetch_pass']:.3f} ms (the FETCH_SIZE pass) = {gbps:.0f} GB/s = {gbps/8000:.2f} of the 8 TB/s HBM peak")
lines.append(f"all kernels of a step: {step_bytes/1e9:.3f} GB through HBM")
extra = ("TCC_HIT_sum", "TCC_MISS_sum", "TCP_TCC_READ_REQ_sum", "TCP_TOTAL_CACHE_ACCESSES_sum", "TCP_TCC_READ_REQ_LATENCY_sum",
         "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SALU", "SQ_INSTS_MFMA", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES",
         "SQ_INST_CYCLES_VMEM", "SQ_INSTS_VALU", "SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES")
lines.append("\n## further counters of the list scans, per dispatch\n")
for k, e in pmc.items():
    if not is_scan(k):
        continue
    parts = []
    for c in extra:
        if c in e:
            grp_n = max(max(v for kk, v in e.items() if kk.startswith("_dispatches_")), 1)
            parts.append(f"{c} {e[c] / grp_n:.4g}")
    if e.get("TCC_HIT_sum", 0) + e.get("TCC_MISS_sum", 0) > 0:
        parts.append(f"L2 hit rate {e['TCC_HIT_sum'] / (e['TCC_HIT_sum'] + e['TCC_MISS_sum']):.3f}")
    lines.append(f"* {k}: " + ", ".join(parts))
out["_hbm_bytes_per_step"] = step_bytes
out["_workload"] = tj["config"]["workload"]
open(os.path.join(here, f"{tag}_summary.md"), "w").write("\n".join(lines) + "\n")
json.dump(out, open(os.path.join(here, f"{tag}_pmc.json"), "w"), indent=1)
print("\n".join(lines))
