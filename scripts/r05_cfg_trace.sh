#!/bin/bash
# kernel timeline of the last search of scripts/bench_configs.py --cfg $1 --nprobes 32 -> gpurun_out/timeline_cfg$1_$2.txt
c=${1:-5}; tag=${2:-x}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
cd $R
out=/tmp/tl_cfg$c; rm -rf $out; mkdir -p $out
timeout 900 rocprofv3 --kernel-trace -d $out -o t -- python3 scripts/bench_configs.py --cfg $c --nprobes 32 --ref-sample 0 --sample 8 > $out/run.log 2>&1
python3 scripts/timeline.py $out/t_results.db last 40 > gpurun_out/timeline_cfg${c}_$tag.txt 2>&1
grep -v "rocclr\|fill_" gpurun_out/timeline_cfg${c}_$tag.txt | tail -45
