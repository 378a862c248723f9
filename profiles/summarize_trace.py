#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV of a bench.py run.

usage: summarize_trace.py <kernel_trace.csv> <bench_stdout_with_json_line> > summary.md

Prints (a) whole-process per-kernel totals for the engine's kernels and (b) the same restricted to
the TIMED region of bench.py: the last steps x launches_per_step dispatches of each engine kernel
(warm-up, trace training and hyper-parameter search run before the timed steps in the same process),
so that the scan kernel's average duration can be compared with roofline.avg_launch_ms."""
import csv
import json
import sys
from collections import defaultdict

trace, benchlog = sys.argv[1], sys.argv[2]
j = None
for line in open(benchlog):
    i = line.find('{"metric"')
    if i >= 0:
        j = json.loads(line[i:])
rows = defaultdict(list)
with open(trace) as f:
    for r in csv.DictReader(f):
        name = r["Kernel_Name"]
        if "amdivf::" not in name:
            continue
        short = name.split("amdivf::")[1].split("(")[0]
        rows[short].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]),
                            r.get("VGPR_Count", ""), r.get("LDS_Block_Size", ""), r.get("Grid_Size", "")))
print("| kernel | calls | total ms | avg ms | min ms | max ms | VGPR | LDS B |")
print("|---|---|---|---|---|---|---|---|")
for k, v in sorted(rows.items(), key=lambda kv: -sum(x[1] for x in kv[1])):
    d = [x[1] / 1e6 for x in v]
    print(f"| {k} | {len(d)} | {sum(d):.3f} | {sum(d)/len(d):.4f} | {min(d):.4f} | {max(d):.4f} | {v[0][2]} | {v[0][3]} |")
if j:
    steps = j["steps"]
    n = int(round(j["roofline"]["launches_per_step"] * steps))
    print()
    print(f"timed region = last {steps} steps; bench.py reported avg_launch_ms = {j['roofline']['avg_launch_ms']:.4f} "
          f"over {n} scan launches, achieved {j['roofline']['achieved']:.0f} GB/s algorithmic")
    sc = sorted(rows.get("scan_tiles_kernel<1>", []) + rows.get("scan_tiles_kernel<0>", []))
    # coarse quantisation also runs the tile kernel (one launch per step, before the list scans): the
    # list-scan launches of a step are the n/steps launches that follow it
    per = n // steps + 1
    tail = sc[-per * steps:]
    scan_only = [x for i, x in enumerate(tail) if i % per != 0]
    d = [x[1] / 1e6 for x in scan_only]
    if d:
        print(f"rocprofv3, same dispatches: {len(d)} list-scan launches, avg {sum(d)/len(d):.4f} ms, total {sum(d):.3f} ms")
    print()
    print("bench line: `" + json.dumps(j)[:2000] + "`")
