#!/usr/bin/env python3
"""kernels of the last search in a rocprofv3 --kernel-trace database (rocpd sqlite): start (ms), duration (ms), gap to the
previous kernel's end (negative: they overlap, i.e. ran on different streams), name"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
rows = list(c.execute("select name, start, end from kernels order by start"))
# the last search starts at the last coarse scan that precedes an init_state_kernel by a few kernels
inits = [i for i, r in enumerate(rows) if "init_state" in r[0]]
if not inits:
    sys.exit("no search in this trace")
# a search in the exact tie regime has two passes (two init_state kernels close together): show both
lo = inits[-1]
if len(inits) > 1 and (rows[inits[-1]][1] - rows[inits[-2]][1]) < 6e6 and len(sys.argv) < 3:
    lo = inits[-2]
lo = max(0, lo - (int(sys.argv[3]) if len(sys.argv) > 3 else 14 if len(sys.argv) > 2 else 6))
t0 = rows[lo][1]
prev = None
for n, s, e in rows[lo:]:
    gap = (s - prev) / 1e6 if prev else 0.0
    print("%8.3f  +%6.3f  gap %7.3f  %s" % ((s - t0) / 1e6, (e - s) / 1e6, gap, n.replace("amdivf::", "").replace("void ", "")[:110]))
    prev = e
