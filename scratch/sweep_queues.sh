#!/bin/bash
for cfg in "4 3" "8 3" "16 3" "8 4" "16 6" "2 3"; do
  set -- $cfg
  GPU_MAX_HW_QUEUES=$1 python bench.py --no-cpu --in-flight $2 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); r=j['roofline']; s=j['one_batch_at_a_time']
print('hw queues $1 inflight $2: value %.0f ms/step %.2f scan_launch_ms %.3f select %.2f | solo %.0f q/s %.2f ms' % (j['value'], j['ms_per_step'], r['avg_launch_ms'], r['other_kernels_ms_per_step']['select'], s['value'], s['ms_per_step']))"
done
