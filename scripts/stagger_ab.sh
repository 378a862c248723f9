# the driver's 20-step window with the first submissions out of phase (bench.py --stagger-ms), no legs
for rep in 1 2 3; do for st in 0 0.2 0.4; do
  python bench.py --steps 20 --warmup 5 --no-cpu --no-legs --stagger-ms $st 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('stagger', $st, 'value', round(j['value']), 'ms/step', round(j['ms_per_step'],3))"
done; done
