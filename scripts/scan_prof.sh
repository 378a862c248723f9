#!/bin/bash
# cycle accounting of the waves of scan_mfma_thr_kernel (experiment build of the engine: see ivf_kernels.hip, AUNCEL_AMD_SCAN_PROF)
#   here:  hipcc ... -DAUNCEL_AMD_SCAN_PROF -c auncel_amd/csrc/ivf_kernels.hip -> auncel_amd/lib/libauncel_amd_prof.so
#   gpurun -- 'bash scripts/scan_prof.sh r04'
tag=${1:-r04}
export AUNCEL_AMD_LIB=$PWD/auncel_amd/lib/libauncel_amd_prof.so AUNCEL_AMD_SCAN_PROF=1
for v in "hinted" "resident:AUNCEL_AMD_RESIDENT_GRIDS=1"; do
  name=${v%%:*}; envs=${v#*:}; [ "$envs" = "$name" ] && envs=
  echo "== $name grids"
  env $envs python bench.py --no-cpu --no-legs --in-flight 1 --steps 6 --warmup 2 2>&1 >/dev/null | grep "scan prof" | tail -4
done > gpurun_out/scan_prof_$tag.txt
cat gpurun_out/scan_prof_$tag.txt
