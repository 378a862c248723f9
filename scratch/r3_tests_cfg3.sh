#!/bin/bash
mkdir -p gpurun_out/r3a
timeout 2000 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r3a/pytest.txt
cat gpurun_out/r3a/pytest.txt
timeout 900 python scripts/bench_configs.py --cfg 3 --ref-sample 200 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if not l.startswith('{'): continue
    j = json.loads(l)
    print('cfg', j['config'], 'nprobe', j['nprobe'], 'qps %.0f' % j['qps'], 'scan %.2f select %.2f coarse %.2f' % (j['scan_ms'], j['select_ms'], j['coarse_ms']), 'cpu==', j['gpu_equals_cpu_on_sample'], 'ref==', (j['reference'] or {}).get('gpu_equals_reference'))
"
