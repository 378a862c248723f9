# which of the legs before it moves the fixed_nprobe_32 leg (bench.py --steps 20 --warmup 5, no CPU side, no other configs)
for skip in latency1 latency1,fp32 latency1,id_ties latency1,fp32,id_ties latency1,one_batch latency1,fp32; do
  AUNCEL_BENCH_SKIP_LEGS=$skip python bench.py --steps 20 --warmup 5 --no-cpu --no-other 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('skipped', '$skip', '| headline', round(j['value']), 'fixed32', round(j['fixed_nprobe_32']['value']))"
done
