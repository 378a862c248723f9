#!/bin/bash
mkdir -p gpurun_out/r3a
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r3a/pytest.txt
cat gpurun_out/r3a/pytest.txt
