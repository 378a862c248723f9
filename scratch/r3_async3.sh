#!/bin/bash
for runner in async threads async; do
timeout 150 python bench.py --no-cpu --no-legs --steps 80 --warmup 8 --in-flight 4 --runner $runner 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); r=j['roofline']; print('runner $runner', 'q/s %.0f ms/step %.3f' % (j['value'], j['ms_per_step']), r['in_flight']['avg_launch_ms'], r['other_kernels_ms_per_step'])"
done
