#!/usr/bin/env python3
"""Per-search chains of the timed region in a rocprofv3 --kernel-trace database (rocpd sqlite): for every hardware queue the
kernels in order with start, duration and the gap to the previous kernel on that queue; then, per kernel name, how long it ran alone
in the trace's first (warm-up, one at a time) part versus in flight.   chain.py t_results.db [from_ms] [to_ms]"""
import collections
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
print("columns:", cols)
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
sel = "select name, start, end" + (", " + qcol if qcol else ", 0") + (", stream_id" if "stream_id" in cols and qcol != "stream_id" else ", 0") + " from kernels order by start"
rows = list(c.execute(sel))
scans = [r for r in rows if "scan_mfma_thr" in r[0]]
t_end = scans[-1][2]
lo = t_end - float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else scans[-30][1]
hi = lo + float(sys.argv[3]) * 1e6 if len(sys.argv) > 3 else lo + 12e6
byq = collections.defaultdict(list)
for n, s, e, q, st in rows:
    if s >= lo and s < hi:
        byq[(q, st)].append((n, s, e))
for q, ks in sorted(byq.items(), key=lambda kv: -len(kv[1])):
    print(f"--- queue {q}: {len(ks)} kernels")
    prev = None
    for n, s, e in ks:
        gap = (s - prev) / 1e6 if prev else 0.0
        print("%8.3f  +%6.3f  gap %7.3f  %s" % ((s - lo) / 1e6, (e - s) / 1e6, gap, n.replace("amdivf::", "").replace("void ", "")[:70]))
        prev = e
