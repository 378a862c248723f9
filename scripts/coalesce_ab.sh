# queued steps served together (option "coalesce") x passes at a time: headline of bench.py, no legs
run() { # in-flight coalesce
  python bench.py --steps 40 --warmup 8 --no-cpu --no-legs --in-flight $1 --coalesce $2 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=j['config']
print('in-flight', $1, 'coalesce', $2, 'value', round(j['value']), 'ms/step', round(j['ms_per_step'],3), c['tickets_and_passes'], 'again', c['queries_searched_again_in_timed_region'])"
}
run 6 1
run 3 2
run 4 2
run 2 4
run 3 1
