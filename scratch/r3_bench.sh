#!/bin/bash
# the headline bench line (4 in flight) without the CPU leg, plus one batch at a time
mkdir -p gpurun_out/r3d
timeout 900 python bench.py --no-cpu --steps 20 --warmup 6 2> gpurun_out/r3d/bench.err | tail -1 > gpurun_out/r3d/bench.json
python - <<'PY'
import json
j = json.loads(open("gpurun_out/r3d/bench.json").read())
r = j["roofline"]; o = j.get("one_batch_at_a_time", {})
print("headline q/s %.0f ms/step %.3f | one batch: %.0f q/s %.3f ms/step | scan %.3f ms x%.1f frac %.3f | select %.3f coarse %.3f | fp32 %s | hints %s | recall %.4f" % (
    j["value"], j["ms_per_step"], o.get("value", 0), o.get("ms_per_step", 0), r["avg_launch_ms"], r["launches_per_step"], r["frac"],
    r["other_kernels_ms_per_step"]["select"], r["other_kernels_ms_per_step"]["coarse"], j.get("fp32_path", {}).get("value"), j["config"]["round_hint"], j["config"]["recall_at_10_mean"]))
PY
tail -3 gpurun_out/r3d/bench.err
