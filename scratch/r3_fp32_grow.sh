#!/bin/bash
export AUNCEL_AMD_NO_BYTES=1
for grow in 12 8 6 4; do
export AUNCEL_AMD_ROUND_GROW=$grow
for fl in 1 4; do
timeout 200 python bench.py --no-cpu --no-legs --steps 24 --warmup 6 --in-flight $fl 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('fp32 grow $grow in-flight $fl', 'q/s %.0f ms/step %.3f' % (j['value'], j['ms_per_step']))"
done; done
