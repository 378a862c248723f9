#!/bin/bash
mkdir -p gpurun_out/r3c
AUNCEL_AMD_SELECT=sorted timeout 600 python scratch/dbg_replay_bench.py 2>&1 | grep "\[replay\]\|\[rounds" > gpurun_out/r3c/replay_sorted_all.txt
# the search rounds of the last (timed) step: the tail
tail -45 gpurun_out/r3c/replay_sorted_all.txt
