#!/bin/bash
export AUNCEL_AMD_DEBUG_TIMING=1
bash scripts/r05_trace.sh redo4 "AUNCEL_AMD_COARSE_TIES=redo" --no-cpu --no-legs --no-other --in-flight 4 --steps 24 --warmup 8
bash scripts/r05_trace.sh def4 "" --no-cpu --no-legs --no-other --in-flight 4 --steps 24 --warmup 8
