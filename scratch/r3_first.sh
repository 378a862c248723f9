#!/bin/bash
# round 3, first GPU visit: DPP primitive, parity of the sorted-array selection, one-batch bench sorted vs heap
mkdir -p gpurun_out/r3a
hipcc --offload-arch=gfx950 -O2 scratch/ubench/dpp_wave_shr.hip -o /tmp/dpp 2>/dev/null && /tmp/dpp > gpurun_out/r3a/dpp.txt 2>&1
cat gpurun_out/r3a/dpp.txt
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r3a/pytest.txt
cat gpurun_out/r3a/pytest.txt
for v in "AUNCEL_AMD_SELECT=sorted" "AUNCEL_AMD_SELECT=heap"; do
  env $v timeout 600 python bench.py --no-cpu --no-legs --steps 10 --warmup 3 --in-flight 1 2> gpurun_out/r3a/bench_$v.err | tail -1 > gpurun_out/r3a/bench_$v.json
  python - <<PY
import json
j=json.loads(open("gpurun_out/r3a/bench_$v.json").read()); r=j['roofline']
print("$v", 'q/s %.0f ms/step %.3f scan avg %.3f x%.0f select %.3f coarse %.3f frac %.3f' % (j['value'], j['ms_per_step'], r['avg_launch_ms'], r['launches_per_step'], r['other_kernels_ms_per_step']['select'], r['other_kernels_ms_per_step']['coarse'], r['frac']))
PY
done
