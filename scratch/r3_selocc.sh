#!/bin/bash
run() {
  timeout 600 python bench.py --no-cpu --no-legs --steps 40 --warmup 8 --in-flight $1 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('$2 in-flight $1', 'q/s %.0f ms/step %.3f' % (j['value'], j['ms_per_step']))"
}
run 4 "min_lds default"
for kb in 40 53 80; do
  export AUNCEL_AMD_SELECT_MIN_LDS_KB=$kb
  run 4 "min_lds ${kb}KB"
  run 1 "min_lds ${kb}KB"
done
