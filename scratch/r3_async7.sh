#!/bin/bash
for w in 24 48; do
timeout 400 python bench.py --runner async --no-cpu --no-legs --warmup $w 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('runner async warmup $w value %.0f ms/step %.3f' % (j['value'], j['ms_per_step']))"
done
timeout 400 python bench.py --runner threads --no-cpu --no-legs --warmup 24 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('runner threads warmup 24 value %.0f ms/step %.3f' % (j['value'], j['ms_per_step']))"
