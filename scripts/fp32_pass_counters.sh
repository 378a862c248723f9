#!/bin/bash
# the fp32 path's threshold passes under counters, rounds growing x 6 (three passes) and x 12 (two): per dispatch of scan_filter_kernel
# the fetched bytes (FETCH_SIZE), the matrix instructions and their busy cycles, the wave-cycle split -- "a pass costs by its pairs"
tag=${1:-r06}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export AUNCEL_AMD_NO_BYTES=1
out=gpurun_out/fp32_passes_$tag.txt; : > $out
for g in 6 12; do
  export AUNCEL_AMD_ROUND_GROW=$g
  for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "TCC_HIT_sum TCC_MISS_sum"; do
    d=/tmp/fp32pmc_$g; rm -rf $d; mkdir -p $d
    rocprofv3 --pmc $grp --output-format csv -d $d -- python3 bench.py --no-cpu --no-legs --in-flight 1 --steps 4 --warmup 2 > $d/run.log 2>&1
    echo "== round_grow $g | $grp (averages over the last 12 dispatches of each kernel: 4 steps x passes)" >> $out
    python3 profiles/pmc_by_kernel.py $d 12 | grep -E "scan_filter_kernel|rescore_kernel|scan_lanes_kernel" >> $out
    rm -rf $d
  done
done
cat $out
