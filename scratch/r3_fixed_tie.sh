#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_coarse_ties.py tests/test_gpu_parity.py -x -q 2>&1 | tail -2
for mode in default final eager; do
  if [ $mode != default ]; then export AUNCEL_AMD_TIE_FIX=$mode; fi
  echo "tie_fix=$mode"; timeout 600 python scratch/perf_scan.py 2>/dev/null | grep "k 100"
done
