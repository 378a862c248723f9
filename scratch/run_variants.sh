#!/bin/bash
# bench.py (one batch at a time, no CPU leg) under a few engine settings: one line each
for v in "" "AUNCEL_AMD_MFMA_CHUNK=128" "AUNCEL_AMD_MFMA_CHUNK=512" "AUNCEL_AMD_MFMA_CHUNK=1024" "AUNCEL_AMD_NO_XCD_CHUNKS=1"; do
  env $v python bench.py --no-cpu --no-legs --steps 10 --warmup 3 --in-flight 1 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('$v', 'q/s %.0f ms/step %.3f scan avg %.3f x%.0f select %.3f coarse %.3f frac %.3f' % (j['value'], j['ms_per_step'], r['avg_launch_ms'], r['launches_per_step'], r['other_kernels_ms_per_step']['select'], r['other_kernels_ms_per_step']['coarse'], r['frac']))"
done
