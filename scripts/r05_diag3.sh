#!/bin/bash
bash scripts/r05_trace.sh redo6 "AUNCEL_AMD_COARSE_TIES=redo" --no-cpu --no-legs --no-other --steps 36 --warmup 12 --runner async
