#!/bin/bash
# scan streams per context x batches in flight (default hardware queues), two passes
for rep in 1 2; do
for cfg in "4 3" "2 3" "2 4" "4 4" "2 5" "2 6"; do
  set -- $cfg
  AUNCEL_AMD_SCAN_STREAMS=$1 python bench.py --no-cpu --in-flight $2 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); s=j['one_batch_at_a_time']
print('streams $1 inflight $2: value %.0f ms/step %.2f | solo %.0f q/s %.2f ms' % (j['value'], j['ms_per_step'], s['value'], s['ms_per_step']))"
done
done
