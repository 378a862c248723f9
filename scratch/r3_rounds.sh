#!/bin/bash
mkdir -p gpurun_out/r3g
for first in 6 8 12 16 24; do for grow in 8 12 24; do
  AUNCEL_AMD_ROUND_FIRST=$first AUNCEL_AMD_ROUND_GROW=$grow timeout 600 python bench.py --no-cpu --steps 20 --warmup 6 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); o=j.get('one_batch_at_a_time',{}); print('first $first grow $grow', 'headline %.0f q/s (%.3f ms/step) one batch %.0f q/s (%.3f ms) recall %.4f' % (j['value'], j['ms_per_step'], o.get('value',0), o.get('ms_per_step',0), j['config']['recall_at_10_mean']))" | tee -a gpurun_out/r3g/round_sweep.txt
done; done
