#!/bin/bash
# kernel timeline of one step of a BASELINE config: bash scripts/timeline_cfg.sh <cfg> <nprobe> <tag>
cfg=${1:-1}; np=${2:-8}; tag=${3:-r04}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
cd $R
out=/tmp/tl_cfg$cfg; rm -rf $out; mkdir -p $out
timeout 900 rocprofv3 --kernel-trace -d $out -o t -- python3 scripts/bench_configs.py --cfg $cfg --nprobes $np --ref-sample 0 --sample 8 > $out/run.log 2>&1
python3 scripts/timeline.py $out/t_results.db last ${4:-14} > gpurun_out/timeline_cfg${cfg}_$tag.txt 2>&1
cat gpurun_out/timeline_cfg${cfg}_$tag.txt
