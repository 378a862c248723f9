#!/bin/bash
# usage: r3_cfg_prof.sh <cfg> <nprobe>
CFG=${1:-5}; NP=${2:-32}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/cfgprof_$CFG
mkdir -p $OUT
cd $R
timeout 900 rocprofv3 --kernel-trace -d $OUT -o t -- python3 scripts/bench_configs.py --cfg $CFG --nprobes $NP --ref-sample 0 --sample 8 > $OUT/run.log 2>&1
grep '^{' $OUT/run.log | cut -c1-400
ls -la $OUT
