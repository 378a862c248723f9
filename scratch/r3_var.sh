#!/bin/bash
for rep in 1 2 3 4 5 6; do
st=60; wu=6
if [ $((rep % 2)) = 0 ]; then st=240; wu=30; fi
timeout 600 python bench.py --no-cpu --no-legs --steps $st --warmup $wu --in-flight 4 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('rep $rep steps $st warmup $wu', 'q/s %.0f ms/step %.3f' % (j['value'], j['ms_per_step']))"
done
