#!/bin/bash
run() {
  timeout 600 python bench.py --no-cpu --no-legs --steps 48 --warmup 8 --in-flight $1 --stagger-ms $3 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('hwq $2 in-flight $1 stagger $3', 'q/s %.0f ms/step %.3f' % (j['value'], j['ms_per_step']))"
}
for q in 4 8; do
  export GPU_MAX_HW_QUEUES=$q
  for fl in 2 3 4 6; do run $fl $q 0.8; done
done
export GPU_MAX_HW_QUEUES=4
run 4 4 0.0
run 4 4 0.4
run 4 4 1.5
