#!/bin/bash
# kernel timeline of one one-query adaptive call on the cfg-2 index (rocprofv3 --kernel-trace): gpurun_out/timeline_batch1_<tag>.txt
tag=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
cd $R
out=/tmp/tl_b1; rm -rf $out; mkdir -p $out
( export ONLY=adaptive CALLS=40; timeout 900 rocprofv3 --kernel-trace -d $out -o t -- python3 scratch/latency1.py > $out/run.log 2>&1 )
tail -3 $out/run.log
python3 scripts/timeline.py $out/t_results.db last > gpurun_out/timeline_batch1_$tag.txt 2>&1
cat gpurun_out/timeline_batch1_$tag.txt
