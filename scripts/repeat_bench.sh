#!/bin/bash
# N plain `python bench.py --steps 20 --warmup 5` runs (the driver's flags) back to back on one box -> gpurun_out/bench_lines_repeat_<tag>.jsonl
# (the first with every leg, the others with the fixed_nprobe_32 leg only: the timed region is the same code either way)
tag=${1:-r06}; n=${2:-10}
out=gpurun_out/bench_lines_repeat_$tag.jsonl
: > $out
python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 >> $out
for i in $(seq 2 $n); do
  AUNCEL_BENCH_SKIP_LEGS=one_batch,fp32,id_ties,latency1 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-other 2>/dev/null | tail -1 >> $out
done
python - <<PY
import json, statistics
v, f = [], []
for l in open("$out"):
    d = json.loads(l)
    v.append(d["value"]); f.append((d.get("fixed_nprobe_32") or {}).get("value"))
    print(round(d["value"]), round(d["ms_per_step"], 3), "fixed32", f[-1] and round(f[-1]))
f = [x for x in f if x]
for name, a in (("headline", v), ("fixed_nprobe_32", f)):
    a = sorted(a)
    print(name, "min", round(a[0]), "median", round(statistics.median(a)), "max", round(a[-1]), "p10/median", round(a[max(0, len(a) // 10)] / statistics.median(a), 3))
PY
