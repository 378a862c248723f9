#!/bin/bash
# in-flight x hardware-queue sweep of the headline configuration
mkdir -p gpurun_out/r3f
for q in 4 8; do for f in 2 4 6 8; do
  GPU_MAX_HW_QUEUES=$q timeout 600 python bench.py --no-cpu --no-legs --steps 24 --warmup 8 --in-flight $f 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('queues $q in-flight $f', 'q/s %.0f ms/step %.3f' % (j['value'], j['ms_per_step']))" | tee -a gpurun_out/r3f/sweep.txt
done; done
