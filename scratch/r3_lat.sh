#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_coarse_ties.py -x -q 2>&1 | tail -2
timeout 600 python scratch/latency1.py 2>/dev/null | tail -5
AUNCEL_AMD_COARSE_TIES=redo timeout 600 python bench.py --no-cpu --no-legs --steps 12 --warmup 4 --in-flight 1 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('redo in-flight 1', 'q/s %.0f ms/step %.3f' % (j['value'], j['ms_per_step']))"
