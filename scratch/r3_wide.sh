#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_filter.py -x -q 2>&1 | tail -8
for nar in 0 1; do
  if [ $nar = 1 ]; then export AUNCEL_AMD_FILTER_NARROW=1; fi
  timeout 600 python scripts/bench_configs.py --cfg 5 --nprobes 32,64 --ref-sample 200 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if not l.startswith('{'): continue
    j = json.loads(l)
    print('narrow=$nar cfg', j['config'], 'nprobe', j['nprobe'], 'qps %.0f' % j['qps'], 'scan %.2f select %.2f coarse %.2f' % (j['scan_ms'], j['select_ms'], j['coarse_ms']), 'cpu==', j['gpu_equals_cpu_on_sample'], 'ref==', (j['reference'] or {}).get('gpu_equals_reference'))
"
done
