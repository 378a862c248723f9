for skip in "" "one_batch" "fp32" "exact_ties" "one_batch,fp32,exact_ties"; do
  AUNCEL_BENCH_SKIP_LEGS=$skip python bench.py --no-cpu --no-other --steps 32 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('skip=[$skip]', round(d['value']), 'async', round(d['single_caller_async']['value']))"
done
