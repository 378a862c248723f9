#!/bin/bash
for sl in 1 2 3 4; do
  AUNCEL_AMD_SLICES=$sl timeout 600 python bench.py --no-cpu --no-legs --steps 20 --warmup 6 --in-flight 1 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('slices $sl in-flight 1', 'q/s %.0f ms/step %.3f recall %.4f' % (j['value'], j['ms_per_step'], j['config']['recall_at_10_mean']))"
done
