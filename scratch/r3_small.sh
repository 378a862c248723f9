#!/bin/bash
for rep in 1 2; do
for g in 1 0; do
export AUNCEL_AMD_SMALL_GATHER=$g
for fl in 1 4; do
timeout 600 python bench.py --no-cpu --no-legs --steps 48 --warmup 8 --in-flight $fl 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('gather $g in-flight $fl', 'q/s %.0f ms/step %.3f' % (j['value'], j['ms_per_step']))"
done; done; done
