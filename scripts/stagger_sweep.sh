#!/bin/bash
# the headline at the driver's flags (--steps 20 --warmup 5) for several start offsets between the caller threads, three runs each,
# interleaved so that drift of the box hits every setting alike -> gpurun_out/stagger_sweep_<tag>.txt
tag=${1:-r04}
out=gpurun_out/stagger_sweep_$tag.txt
: > $out
for rep in 1 2 3; do
  for st in ${STAGGERS:-0 0.4 0.8 1.2 1.7}; do
    v=$(BENCH_STAGGER_MS=$st python bench.py --gpus 1 --steps 20 --warmup 5 --no-legs --no-other 2>/dev/null | tail -1 |
        python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), d['ms_per_step'])")
    echo "stagger_ms $st rep $rep: $v" | tee -a $out
  done
done
