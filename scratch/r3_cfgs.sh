#!/bin/bash
mkdir -p gpurun_out/r3e
for f in 1 0; do
  AUNCEL_AMD_FILTER=$f timeout 1500 python scripts/bench_configs.py --cfg 5,3 --ref-sample 500 2>gpurun_out/r3e/cfg_f$f.err | python -c "
import sys, json
for l in sys.stdin:
    if not l.startswith('{'): continue
    j = json.loads(l)
    print('filter=$f cfg', j['config'], 'nprobe', j['nprobe'], 'qps %.0f' % j['qps'], 'scan %.2f select %.2f coarse %.2f' % (j['scan_ms'], j['select_ms'], j['coarse_ms']), 'cpu==', j['gpu_equals_cpu_on_sample'], 'ref==', (j['reference'] or {}).get('gpu_equals_reference'), 'recall %.4f' % j['recall_at_k'])
" | tee -a gpurun_out/r3e/cfgs.txt
done
tail -3 gpurun_out/r3e/cfg_f1.err
