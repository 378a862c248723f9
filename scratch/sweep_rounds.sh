#!/bin/bash
# sweep of the round schedule / batches in flight on the bench workload: prints one line per setting
for cfg in "12 3.5 3" "12 2.5 3" "12 2.0 3" "8 2.5 3" "8 2.0 3" "6 2.0 3" "16 3.5 3" "12 3.5 4" "12 2.5 4" "12 3.5 2" "12 3.5 6"; do
  set -- $cfg
  AUNCEL_AMD_ROUND_FIRST=$1 AUNCEL_AMD_ROUND_GROW=$2 python bench.py --no-cpu --in-flight $3 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); r=j['roofline']; s=j['one_batch_at_a_time']
print('first $1 grow $2 inflight $3: value %.0f ms/step %.2f rounds %.1f overscan %.2f scan_launch_ms %.3f select %.2f | solo %.0f q/s %.2f ms' % (j['value'], j['ms_per_step'], r['launches_per_step'], r['computed_over_algorithmic'], r['avg_launch_ms'], r['other_kernels_ms_per_step']['select'], s['value'], s['ms_per_step']))"
done
