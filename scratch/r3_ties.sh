#!/bin/bash
timeout 1500 python -m pytest tests/test_gpu_coarse_ties.py tests/test_gpu_parity.py tests/test_host_mirror.py -m gpu -x -q 2>&1 | tail -6
for t in "" "redo"; do
  AUNCEL_AMD_COARSE_TIES=$t timeout 900 python bench.py --no-cpu --no-legs --steps 20 --warmup 6 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('ties=$t', 'q/s %.0f ms/step %.3f' % (j['value'], j['ms_per_step']))"
done
AUNCEL_AMD_COARSE_TIES=redo timeout 900 python bench.py --no-cpu --no-legs --steps 24 --warmup 8 --in-flight 8 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('ties=redo in-flight 8', 'q/s %.0f ms/step %.3f' % (j['value'], j['ms_per_step']))"
