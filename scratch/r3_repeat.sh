#!/bin/bash
rm -f gpurun_out/bench_repeat_r03.jsonl
for rep in 1 2 3; do
timeout 400 python bench.py 2>/dev/null | tail -1 >> gpurun_out/bench_repeat_r03.jsonl
done
python - <<'PY'
import json
for l in open('gpurun_out/bench_repeat_r03.jsonl'):
    j=json.loads(l); print('value %.0f one-batch %.3f single %.0f fp32 %.0f exact %.0f' % (j['value'], j['one_batch_at_a_time']['ms_per_step'], j['single_caller_async']['value'], j['fp32_path']['value'], j['exact_tie_order']['value']))
PY
