#!/bin/bash
# Kernel timelines (rocprofv3 --kernel-trace) of one step of three side legs, as text under gpurun_out/:
#   fp32_path (headline workload, byte codes off), the headline itself (exact tie regime: the heap beside round 0, the patch before its
#   selection), the same with runs in centroid-number order, cfg 5 (d = 960, nprobe 32)
tag=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
cd $R
one() {  # name, env assignment, command...
  name=$1; shift; envs=$1; shift
  out=/tmp/tl_$name; rm -rf $out; mkdir -p $out
  ( export $envs; timeout 900 rocprofv3 --kernel-trace -d $out -o t -- "$@" > $out/run.log 2>&1 )
  python3 scripts/timeline.py $out/t_results.db > gpurun_out/timeline_${name}_$tag.txt 2>&1
  rm -rf $out
}
one fp32_path AUNCEL_AMD_NO_BYTES=1 python3 bench.py --no-cpu --no-legs --in-flight 1 --steps 6 --warmup 3
one exact_ties AUNCEL_AMD_X=0 python3 bench.py --no-cpu --no-legs --in-flight 1 --steps 6 --warmup 3
one id_ties AUNCEL_AMD_X=0 python3 bench.py --no-cpu --no-legs --in-flight 1 --steps 6 --warmup 3 --coarse-ties id
one cfg5 AUNCEL_AMD_X=0 python3 scripts/bench_configs.py --cfg 5 --nprobes 32 --ref-sample 0 --sample 8
wc -l gpurun_out/timeline_*_$tag.txt
