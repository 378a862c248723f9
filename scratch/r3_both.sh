#!/bin/bash
# parity suite, then the one-batch kernel trace (default build, and the 4-waves-per-SIMD variant of the dense selection)
mkdir -p gpurun_out/r3a
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r3a/pytest.txt
cat gpurun_out/r3a/pytest.txt
bash scratch/r3_trace.sh 2>&1 | grep -v "PlanArgs\|copyBuffer\|fillBuffer\|plan_reset" | tail -40
if [ -f auncel_amd/lib/libauncel_amd_w4.so ]; then
  echo "== variant w4"
  AUNCEL_AMD_LIB=$PWD/auncel_amd/lib/libauncel_amd_w4.so bash scratch/r3_trace.sh 2>&1 | grep "us/step" | head -4
fi
