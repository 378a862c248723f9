#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=/tmp/busyprof; rm -rf $OUT; mkdir -p $OUT
cd $R
timeout 900 rocprofv3 --kernel-trace -d $OUT -o t -- python3 bench.py --no-cpu --no-legs --steps 48 --warmup 12 > $OUT/run.log 2>&1
tail -1 $OUT/run.log | cut -c1-120
python3 scripts/busy.py $OUT/t_results.db 60 10
