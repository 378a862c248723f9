#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_random_adaptive.py tests/test_gpu_filter.py -x -q 2>&1 | tail -2
timeout 400 python bench.py --no-cpu --steps 40 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('value', j['value'], 'fp32', j['fp32_path'], 'exact', j['exact_tie_order']['value'], 'single', j['single_caller_async']['value'])"
