#!/bin/bash
# threshold-round selection with and without the rows' candidate lists: per-wave counters (AUNCEL_AMD_DEBUG_REPLAY) and the kernel
# timeline of one step, one batch in flight -> gpurun_out/rowlists_<tag>.txt
tag=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
cd $R
o=gpurun_out/rowlists_$tag.txt
: > $o
for rl in 0 1; do
  echo "== row_lists $rl: per-wave counters of the last threshold-round selection" >> $o
  AUNCEL_AMD_ROW_LISTS=$rl AUNCEL_AMD_DEBUG_REPLAY=1 python3 bench.py --in-flight 1 --steps 2 --warmup 1 --no-legs --no-other --no-cpu 2>&1 | grep "\[replay\]" | tail -9 >> $o
  out=/tmp/tl_rl$rl; rm -rf $out; mkdir -p $out
  ( export AUNCEL_AMD_ROW_LISTS=$rl; timeout 900 rocprofv3 --kernel-trace -d $out -o t -- python3 bench.py --no-cpu --no-legs --no-other --in-flight 1 --steps 6 --warmup 3 > $out/run.log 2>&1 )
  echo "== row_lists $rl: timeline of the last step" >> $o
  python3 scripts/timeline.py $out/t_results.db >> $o 2>&1
  rm -rf $out
done
cat $o
