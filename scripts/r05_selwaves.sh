#!/bin/bash
# selection kernel residency (AUNCEL_SEL_WAVES waves a SIMD asked for) against the headline
cd "$GRAFT_REPO_ROOT"
for v in 5 4 6 3; do
  export AUNCEL_AMD_CXXFLAGS="-DAUNCEL_SEL_WAVES=$v"
  python -c "from auncel_amd import build as b; b.build()" > /dev/null 2>&1 || { echo "SEL_WAVES $v: build failed"; continue; }
  for rep in 1 2; do
    python bench.py --no-cpu --no-legs --no-other 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('SEL_WAVES $v rep $rep: %.3f M q/s, %.3f ms/step' % (d['value'] / 1e6, d['ms_per_step']))"
  done
done
unset AUNCEL_AMD_CXXFLAGS
