#!/bin/bash
# bench.py (default: four batches in flight, no CPU leg) under a few engine settings: one line each
for v in "AUNCEL_AMD_ROUND_GROW=12" "AUNCEL_AMD_ROUND_GROW=3.5" "AUNCEL_AMD_ROUND_GROW=6" "AUNCEL_AMD_SYNC_ROUNDS=1 AUNCEL_AMD_ROUND_GROW=3.5"; do
  env $v python bench.py --no-cpu --steps 24 --warmup 4 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']; o=j['one_batch_at_a_time']
print('$v', 'q/s %.0f ms/step %.3f | one batch: q/s %.0f ms/step %.3f scan avg %.3f select %.3f | scan x%.0f over-scan %.2f' % (j['value'], j['ms_per_step'], o['value'], o['ms_per_step'], o['scan_avg_launch_ms'], o['other_kernels_ms_per_step']['select'], r['launches_per_step'], r['computed_over_algorithmic']))"
done
