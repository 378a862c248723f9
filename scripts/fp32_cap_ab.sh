for rep in 1 2 3; do
for v in nocap cap; do
  if [ $v = cap ]; then unset AUNCEL_AMD_NO_POOL_FP32_CAP; else export AUNCEL_AMD_NO_POOL_FP32_CAP=1; fi
  AUNCEL_BENCH_FP32_STEPS=24 AUNCEL_BENCH_SKIP_LEGS=one_batch,id_ties,fixed,latency1 python bench.py --steps 6 --warmup 6 --no-cpu --no-other 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); f=j['fp32_path']
print('$v', 'headline', round(j['value']), 'fp32', round(f['value']), 'ms', round(f['ms_per_step'],3), 'alone', round(f['one_batch_at_a_time_ms'],3))"
done; done
