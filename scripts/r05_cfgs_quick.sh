#!/bin/bash
# cfg 5 and 3 at nprobe 32: throughput, phases, what each round planned (gpurun_out/cfgs_quick.txt)
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/cfgs_quick.txt; : > $out
for c in ${CFGS:-5 3}; do
  AUNCEL_AMD_DEBUG_ROUNDS=1 python scripts/bench_configs.py --cfg $c --nprobes ${NPROBES:-32} --ref-sample 0 --sample 32 2>gpurun_out/err.txt | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); ph = d['phases']
        print('cfg', d['config'], 'nprobe', d['nprobe'], 'q/s %.3fM' % (d['value'] / 1e6), 'same', d['gpu_equals_cpu_on_sample'], {k: round(v['ms'], 3) for k, v in ph.items()})
" >> $out
  grep "rounds\]" gpurun_out/err.txt | tail -2 >> $out
done
cat $out
