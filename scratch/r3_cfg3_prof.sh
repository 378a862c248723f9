#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/cfg3prof
cd $R
rocprofv3 --kernel-trace --stats -d gpurun_out/cfg3prof -o t -- python3 scripts/bench_configs.py --cfg 3 --nprobes 16 --ref-sample 0 --sample 8 > gpurun_out/cfg3prof/run.log 2>&1
tail -2 gpurun_out/cfg3prof/run.log | cut -c1-600
f=$(find gpurun_out/cfg3prof -name "*kernel_stats.csv" | head -1)
head -25 $f | cut -c1-200
