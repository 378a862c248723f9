#!/bin/bash
# experiment: is the dense round of cfg 5 bound by its scalar query-operand loads (30 KB a query group, past the scalar cache)?
cd "$GRAFT_REPO_ROOT"
for f in "" "-DAUNCEL_SCAN_QHACK=1"; do
  export AUNCEL_AMD_CXXFLAGS="$f"
  python -c "from auncel_amd import build as b; b.build()" > /dev/null 2>&1
  python scripts/bench_configs.py --cfg 5 --nprobes 32 --ref-sample 0 --sample 8 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); ph = d['phases']
        print('flags [$f] cfg', d['config'], 'q/s %.3fM' % (d['value'] / 1e6), 'dense %.3f ms' % ph['scan_dense']['ms'], 'thr %.3f' % ph['scan_thr']['ms'], 'same', d['gpu_equals_cpu_on_sample'])
"
done
