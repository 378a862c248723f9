#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/fp32prof
rm -rf $OUT; mkdir -p $OUT
cd $R
export AUNCEL_AMD_NO_BYTES=1
timeout 900 rocprofv3 --kernel-trace -d $OUT -o t -- python3 bench.py --no-cpu --no-legs --in-flight 1 --steps 8 --warmup 4 > $OUT/run.log 2>&1
tail -1 $OUT/run.log | cut -c1-300
ls -la $OUT | head
