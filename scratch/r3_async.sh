#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "asynchronous or page_locked" 2>&1 | tail -3
for runner in async threads async threads; do
timeout 600 python bench.py --no-cpu --no-legs --steps 60 --warmup 8 --in-flight 4 --runner $runner 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('runner $runner in-flight 4', 'q/s %.0f ms/step %.3f recall %.4f' % (j['value'], j['ms_per_step'], j['config']['recall_at_10_mean']), j['config']['round_hint'])"
done
timeout 900 python bench.py --no-cpu --steps 30 2>gpurun_out/async_err.txt | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('full line:', j['value'], j['ms_per_step'], 'one batch', j['one_batch_at_a_time']['ms_per_step'], 'fp32', j['fp32_path']['value'], j['fp32_path']['same_results_as_byte_codes'], 'exact', j['exact_tie_order'], j['config']['runner'])"
tail -3 gpurun_out/async_err.txt
