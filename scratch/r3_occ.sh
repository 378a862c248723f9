#!/bin/bash
run() {
  timeout 600 python bench.py --no-cpu --no-legs --steps 40 --warmup 8 --in-flight $1 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('$2 in-flight $1', 'q/s %.0f ms/step %.3f' % (j['value'], j['ms_per_step']))"
}
run 4 default
run 1 default
for pc in 2 3 4; do
  export AUNCEL_AMD_RESIDENT_GRIDS=1 AUNCEL_AMD_MFMA_WG_PER_CU=$pc
  run 4 "resident wg_per_cu=$pc"
  run 1 "resident wg_per_cu=$pc"
done
