#!/bin/bash
# round schedule of the byte-code path: first round (dense) x growth, four batches in flight + one at a time
for v in "AUNCEL_AMD_ROUND_FIRST=12 AUNCEL_AMD_ROUND_GROW=12" "AUNCEL_AMD_ROUND_FIRST=8 AUNCEL_AMD_ROUND_GROW=18" "AUNCEL_AMD_ROUND_FIRST=6 AUNCEL_AMD_ROUND_GROW=24" "AUNCEL_AMD_ROUND_FIRST=4 AUNCEL_AMD_ROUND_GROW=36" "AUNCEL_AMD_ROUND_FIRST=16 AUNCEL_AMD_ROUND_GROW=9" "AUNCEL_AMD_ROUND_FIRST=24 AUNCEL_AMD_ROUND_GROW=6" "AUNCEL_AMD_ROUND_FIRST=8 AUNCEL_AMD_ROUND_GROW=8"; do
  env $v python bench.py --no-cpu --steps 24 --warmup 4 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']; o=j['one_batch_at_a_time']
print('$v', 'q/s %.0f ms/step %.3f | one batch: q/s %.0f ms/step %.3f scan avg %.3f select %.3f | scan x%.0f over-scan %.2f' % (j['value'], j['ms_per_step'], o['value'], o['ms_per_step'], o['scan_avg_launch_ms'], o['other_kernels_ms_per_step']['select'], r['launches_per_step'], r['computed_over_algorithmic']))"
done
