#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=/tmp/redoprof
rm -rf $OUT; mkdir -p $OUT $R/gpurun_out/redoprof
cd $R
export AUNCEL_AMD_COARSE_TIES=redo
timeout 900 rocprofv3 --kernel-trace -d $OUT -o t -- python3 bench.py --no-cpu --no-legs --in-flight 1 --steps 6 --warmup 3 > $OUT/run.log 2>&1
tail -1 $OUT/run.log | cut -c1-200
python3 scratch/timeline_steps.py $OUT/t_results.db > $R/gpurun_out/redoprof/timeline.txt
tail -150 $R/gpurun_out/redoprof/timeline.txt
