#!/bin/bash
t0=$(date +%s)
timeout 120 python -m pytest tests/test_gpu_parity.py -x -q -k "asynchronous" 2>&1 | tail -3
echo "pytest wall $(( $(date +%s) - t0 )) s, rc ${PIPESTATUS[0]}"
t0=$(date +%s)
timeout 200 python bench.py --no-cpu --no-legs --steps 24 --warmup 8 --in-flight 4 --runner async 2>gpurun_out/async_err.txt | tail -1 | cut -c1-230
echo "bench wall $(( $(date +%s) - t0 )) s"
grep -v multipler gpurun_out/async_err.txt | tail -4 | cut -c1-200
ps aux | grep -c "[p]ython"
