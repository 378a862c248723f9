#!/bin/bash
mkdir -p gpurun_out/r3a
timeout 1500 python -m pytest tests -m gpu -x -q -k "not bench_modes and not integration and not host_mirror" 2>&1 | tail -4
AUNCEL_AMD_FILTER=1 timeout 900 python scripts/bench_configs.py --cfg 5,3 --ref-sample 200 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if not l.startswith('{'): continue
    j = json.loads(l)
    print('cfg', j['config'], 'nprobe', j['nprobe'], 'qps %.0f' % j['qps'], 'scan %.2f select %.2f coarse %.2f' % (j['scan_ms'], j['select_ms'], j['coarse_ms']), 'cpu==', j['gpu_equals_cpu_on_sample'], 'ref==', (j['reference'] or {}).get('gpu_equals_reference'))
"
