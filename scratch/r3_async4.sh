#!/bin/bash
timeout 400 python bench.py 2>gpurun_out/async_err.txt | tail -1 > gpurun_out/async_line.json
python -c "
import json
j=json.load(open('gpurun_out/async_line.json'))
print('value', j['value'], j['ms_per_step'], j['config']['runner'])
print('single caller', j.get('single_caller_async'))
print('one batch', j['one_batch_at_a_time']['ms_per_step'])
print('fp32', j['fp32_path']['value'], j['fp32_path']['same_results_as_byte_codes'])
print('exact', j['exact_tie_order'])
print('parity', j['cpu_baseline']['gpu_matches_cpu_on_sample'], j['cpu_baseline']['parity']['timed_configuration_queries_differing'], j['cpu_baseline']['value'])
"
grep -v multipler gpurun_out/async_err.txt | tail -3 | cut -c1-250
