#!/bin/bash
export AUNCEL_AMD_NO_BYTES=1
for inc in 6 12 18 24; do
export AUNCEL_AMD_ROUND_INC=$inc
timeout 200 python bench.py --no-cpu --no-legs --steps 24 --warmup 6 --in-flight 4 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('fp32 inc $inc in-flight 4', 'q/s %.0f ms/step %.3f' % (j['value'], j['ms_per_step']))"
done
