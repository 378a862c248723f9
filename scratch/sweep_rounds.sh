#!/bin/bash
# sweep of the round schedule / batches in flight on the bench workload: prints one line per setting
# columns: first-round probes, minimum increment of later rounds, growth, batches in flight
for cfg in ${SWEEP:-"12 12 3.5 3" "1 12 3.5 3" "2 12 3.5 3" "4 12 3.5 3" "1 16 3.5 3" "1 24 3.5 3" "1 12 5 3" "1 12 3.5 4"}; do
  set -- $cfg
  AUNCEL_AMD_ROUND_FIRST=$1 AUNCEL_AMD_ROUND_INC=$2 AUNCEL_AMD_ROUND_GROW=$3 python bench.py --no-cpu --in-flight $4 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); r=j['roofline']; s=j['one_batch_at_a_time']
print('first $1 inc $2 grow $3 inflight $4: value %.0f ms/step %.2f rounds %.1f overscan %.2f scan_launch_ms %.3f select %.2f | solo %.0f q/s %.2f ms' % (j['value'], j['ms_per_step'], r['launches_per_step'], r['computed_over_algorithmic'], r['avg_launch_ms'], r['other_kernels_ms_per_step']['select'], s['value'], s['ms_per_step']))"
done
