# the fp32 path's round growth (option round_grow; 6 by default for fp32 searches with the filter): fp32_path of bench.py, 24-step leg
for g in 0 8 12 0 8 12; do
  if [ $g = 0 ]; then unset AUNCEL_AMD_ROUND_GROW; else export AUNCEL_AMD_ROUND_GROW=$g; fi
  AUNCEL_BENCH_FP32_STEPS=24 AUNCEL_BENCH_SKIP_LEGS=one_batch,id_ties,fixed,latency1 python bench.py --steps 6 --warmup 6 --no-cpu --no-other 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); f=j['fp32_path']
print('grow', '$g', 'fp32', round(f['value']), 'ms', round(f['ms_per_step'],3), 'alone', round(f['one_batch_at_a_time_ms'],3), {k: round(v,3) for k,v in f['roofline']['phases_ms_per_step'].items()})"
done
