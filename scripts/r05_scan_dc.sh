#!/bin/bash
# the dense round of float data by staging chunk (AUNCEL_SCAN_DC dimensions a row) and fetch lead (AUNCEL_SCAN_PF chunks): cfg 5 / cfg 3
# at nprobe 32, one line each -> gpurun_out/scan_dc.txt
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/scan_dc.txt; : > $out
for v in "16 2" "32 1" "32 2" "64 1"; do
  set -- $v
  export AUNCEL_AMD_CXXFLAGS="-DAUNCEL_SCAN_DC=$1 -DAUNCEL_SCAN_PF=$2"
  python -c "from auncel_amd import build as b; b.build()" > /dev/null 2>&1 || { echo "DC $1 PF $2: build failed" >> $out; continue; }
  for c in ${CFGS:-5 3}; do
    python scripts/bench_configs.py --cfg $c --nprobes 32 --ref-sample 0 --sample 16 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); ph = d['phases']
        print('DC $1 PF $2 cfg', d['config'], 'q/s %.3fM' % (d['value'] / 1e6), 'dense %.3f ms' % ph['scan_dense']['ms'], 'thr %.3f' % ph['scan_thr']['ms'], 'same', d['gpu_equals_cpu_on_sample'])
" >> $out
  done
done
unset AUNCEL_AMD_CXXFLAGS
cat $out
