#!/bin/bash
# cfg 5 / cfg 3 lines (no reference run) + kernel timeline of cfg 5
python3 scripts/bench_configs.py --cfg 5,3 --nprobes 32 --ref-sample 0 --sample 64 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l)
    print('cfg',d['config'],'nprobe',d['nprobe'],'%.3f M q/s'%(d['value']/1e6),'%.2f ms'%d['ms_per_batch'],'== cpu',d['gpu_equals_cpu_on_sample'],{k:round(v['ms'],3) for k,v in d['phases'].items()})
"
