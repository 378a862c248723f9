#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_coarse_ties.py -x -q 2>&1 | tail -4
for fl in 1 4; do
for ties in id redo; do
  AUNCEL_AMD_COARSE_TIES=$ties timeout 600 python bench.py --no-cpu --no-legs --steps 24 --warmup 6 --in-flight $fl 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); c=j['config']
print('ties $ties in-flight $fl', 'q/s %.0f ms/step %.3f recall %.4f' % (j['value'], j['ms_per_step'], c['recall_at_10_mean']), {k: v for k, v in c.items() if 'tie' in k or 'again' in k})"
done; done
