#!/bin/bash
# every kernel of the one-query adaptive calls of scratch/latency1.py (rocprofv3 --kernel-trace), one line each:
# gpurun_out/latency1_calls_<tag>.txt (start ms, duration ms, gap to the previous kernel's end, name)
tag=${1:-r05}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
cd $R
out=/tmp/tl_b1c; rm -rf $out; mkdir -p $out
( export ONLY=adaptive CALLS=${CALLS:-60}; timeout 900 rocprofv3 --kernel-trace -d $out -o t -- python3 scratch/latency1.py > $out/run.log 2>&1 )
tail -2 $out/run.log
python3 - $out/t_results.db > gpurun_out/latency1_calls_$tag.txt <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = list(c.execute("select name, start, end from kernels order by start"))
rows = rows[-6000:]
t0 = rows[0][1]; prev = None
for n, s, e in rows:
    gap = (s - prev) / 1e6 if prev else 0.0
    print("%10.3f +%6.3f gap %7.3f %s" % ((s - t0) / 1e6, (e - s) / 1e6, gap, n.replace("amdivf::", "").replace("void ", "")[:90]))
    prev = e
PY
wc -l gpurun_out/latency1_calls_$tag.txt
