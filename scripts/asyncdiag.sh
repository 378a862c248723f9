# diagnosis helper: a few bench.py runs on one box (edit as needed)
for a in "--no-other"; do
python bench.py $a 2> gpurun_out/asyncdiag.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), d['ms_per_step'], 'async', round((d.get('single_caller_async') or {}).get('value', 0)), 'one', d['one_batch_at_a_time']['ms_per_step'], 'fp32', round((d.get('fp32_path') or {}).get('value', 0)), (d.get('fp32_path') or {}).get('same_results_as_byte_codes'), 'exact', round((d.get('exact_tie_order') or {}).get('value', 0)), 'cpu', (d.get('cpu_baseline') or {}).get('gpu_matches_cpu_on_sample'), ((d.get('cpu_baseline') or {}).get('parity') or {}).get('timed_configuration_queries_differing'))"
done
