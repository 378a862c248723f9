#!/bin/bash
# three plain `python bench.py --steps 20 --warmup 5` runs (the driver's flags) back to back on one box -> gpurun_out/bench_lines_repeat_<tag>.jsonl
tag=${1:-r04}
: > gpurun_out/bench_lines_repeat_$tag.jsonl
for i in 1 2 3; do
  python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 >> gpurun_out/bench_lines_repeat_$tag.jsonl
done
python - <<PY
import json
for l in open("gpurun_out/bench_lines_repeat_$tag.jsonl"):
    d = json.loads(l)
    print(round(d["value"]), d["ms_per_step"], "one-batch", d["one_batch_at_a_time"]["ms_per_step"], "fp32", round(d["fp32_path"]["value"]),
          "async", round(d["single_caller_async"]["value"]), "cfgs", [round(c["value"]) for c in d["other_configs"]],
          "parity", d["cpu_baseline"]["gpu_matches_cpu_on_sample"], d["cpu_baseline"]["parity"]["timed_configuration_queries_differing"])
PY
