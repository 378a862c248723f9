#!/bin/bash
# parity suite, then the one-batch kernel trace
mkdir -p gpurun_out/r3a
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r3a/pytest.txt
cat gpurun_out/r3a/pytest.txt
bash scratch/r3_trace.sh 2>&1 | tail -60
