#!/bin/bash
# round 5, first look: what the exact tie regime costs with several searches in flight (A/B lines + a kernel trace of the region)
bash scripts/ab.sh r05a < scripts/ab_r05a.cfg
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=/tmp/busyprof; rm -rf $OUT; mkdir -p $OUT
cd $R
export AUNCEL_AMD_COARSE_TIES=redo
timeout 900 rocprofv3 --kernel-trace -d $OUT -o t -- python3 bench.py --no-cpu --no-legs --no-other --in-flight 4 --steps 40 --warmup 8 > $OUT/run.log 2>&1
tail -1 $OUT/run.log | cut -c1-120
python3 scripts/busy.py $OUT/t_results.db 60 10 | tee gpurun_out/r05_busy_redo.txt
