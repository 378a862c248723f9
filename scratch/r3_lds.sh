#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_filter.py tests/test_gpu_random.py -x -q 2>&1 | tail -2
bash scratch/r3_fp32_prof.sh > /dev/null 2>&1
python3 scratch/timeline.py gpurun_out/fp32prof/t_results.db 1 0 | grep "scan_filter\|rescore" | head -4
rm -f gpurun_out/fp32prof/t_results.db
AUNCEL_AMD_NO_BYTES=1 timeout 600 python bench.py --no-cpu --no-legs --steps 12 --warmup 4 --in-flight 4 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('fp32 in-flight 4', 'q/s %.0f ms/step %.3f' % (j['value'], j['ms_per_step']))"
timeout 900 python scripts/bench_configs.py --cfg 5,3 --ref-sample 200 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if not l.startswith('{'): continue
    j = json.loads(l)
    print('cfg', j['config'], 'nprobe', j['nprobe'], 'qps %.0f' % j['qps'], 'scan %.2f select %.2f coarse %.2f' % (j['scan_ms'], j['select_ms'], j['coarse_ms']), 'cpu==', j['gpu_equals_cpu_on_sample'], 'ref==', (j['reference'] or {}).get('gpu_equals_reference'))
"
