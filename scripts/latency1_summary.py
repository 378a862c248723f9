#!/usr/bin/env python3
"""per-call summary of gpurun_out/latency1_calls_<tag>.txt (scripts/latency1_calls.sh): for every one-query adaptive call the
span of its kernels and the duration of the selections and of the tie replay in it"""
import re
import sys

rows = []
for l in open(sys.argv[1]):
    m = re.match(r'\s*([\d.]+) \+\s*([\d.]+) gap\s+(-?[\d.]+) (.*)', l)
    if m:
        rows.append((float(m.group(1)), float(m.group(2)), float(m.group(3)), m.group(4)))
starts = [i for i, r in enumerate(rows) if r[3].startswith('small_state_kernel')]
calls = []
for a, b in zip(starts, starts[1:] + [len(rows)]):
    seg = rows[max(a - 4, 0):b - 4]
    t0 = seg[0][0]
    t1 = max(r[0] + r[1] for r in seg)
    sel = [r[1] for r in seg if r[3].startswith('select_sorted')]
    tf = [r[1] for r in seg if r[3].startswith('tie_fix')]
    heap = [r[1] for r in seg if r[3].startswith('heap_tie')]
    calls.append((t1 - t0, sel, tf, heap, len(seg)))
calls.sort(key=lambda c: c[0])
n = len(calls)
print("calls", n, "span median %.3f p90 %.3f max %.3f" % (calls[n // 2][0], calls[int(n * 0.9)][0], calls[-1][0]))
for c in calls:
    print("%.3f  kernels %2d  select %s  tie_fix %s %s" % (c[0], c[4], " ".join("%.3f" % v for v in c[1]), " ".join("%.3f" % v for v in c[2]),
                                                           ("heap %.3f" % c[3][0]) if c[3] else ""))
