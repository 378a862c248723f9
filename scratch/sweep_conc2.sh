#!/bin/bash
# batches in flight x hardware queues (chained rounds, hinted grids)
for q in 4 8; do for f in 3 4 5 6 8; do
  GPU_MAX_HW_QUEUES=$q python bench.py --no-cpu --no-legs --steps 48 --warmup 8 --in-flight $f 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('hw queues $q in flight $f: q/s %.0f ms/step %.3f' % (j['value'], j['ms_per_step']))"
done; done
for st in 0 0.4 1.2; do python bench.py --no-cpu --no-legs --steps 48 --warmup 8 --stagger-ms $st 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('stagger $st ms (4 in flight, 4 queues): q/s %.0f ms/step %.3f' % (j['value'], j['ms_per_step']))"; done
