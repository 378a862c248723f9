#!/bin/bash
# kernel trace (rocprofv3 --kernel-trace) of a bench.py region, the database copied back for analysis:
#   bash scripts/r05_trace.sh <name> "<ENV=VAL ...>" <bench.py arguments...>
name=$1; shift; envs=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=/tmp/trace_$name; rm -rf $OUT; mkdir -p $OUT
cd $R
( [ -n "$envs" ] && export $envs; timeout 900 rocprofv3 --kernel-trace -d $OUT -o t -- python3 bench.py "$@" > $OUT/run.log 2> $OUT/run.err )
tail -1 $OUT/run.log | cut -c1-160
grep "\[host\]" $OUT/run.err | tail -40 > gpurun_out/trace_${name}_host.txt
ls -la $OUT/*.db
python3 - $OUT/t_results.db gpurun_out/trace_$name.npz <<'PY'
import sqlite3, sys, numpy as np
c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
print(cols)
want = [x for x in ("name", "start", "end", "queue_id", "stream_id", "grid_x", "workgroup_x", "grid_size_x", "workgroup_size_x") if x in cols]
rows = list(c.execute("select %s from kernels order by start" % ", ".join(want)))
names = sorted(set(r[0] for r in rows))
idx = {n: i for i, n in enumerate(names)}
arr = {"names": np.array(names)}
arr["name_id"] = np.array([idx[r[0]] for r in rows], dtype=np.int32)
for j, w in enumerate(want[1:], 1):
    arr[w] = np.array([r[j] if r[j] is not None else -1 for r in rows], dtype=np.int64)
np.savez_compressed(sys.argv[2], **arr)
print(len(rows), "kernels")
PY
