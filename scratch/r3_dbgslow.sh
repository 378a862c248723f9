#!/bin/bash
AUNCEL_AMD_DEBUG_TIMING=1 timeout 300 python bench.py --no-cpu --no-legs --steps 6 --warmup 4 --in-flight 2 --runner threads 2>gpurun_out/slow_err.txt | tail -1 | cut -c1-200
grep -v "multipler" gpurun_out/slow_err.txt | tail -40 | cut -c1-220
