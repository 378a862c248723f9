#!/bin/bash
# the streaming hint on scan_lanes_kernel's block loads: always (1), only where one wave reads a block (2), never (0)
cd "$GRAFT_REPO_ROOT"
for v in 1 2 0; do
  export AUNCEL_AMD_CXXFLAGS="-DAUNCEL_LANES_NT=$v"
  python -c "from auncel_amd import build as b; b.build()" > /dev/null 2>&1 || { echo "NT $v: build failed"; continue; }
  echo "== AUNCEL_LANES_NT=$v"
  bash scripts/r05_cfgs_quick.sh | grep -v rounds | cut -c1-105
  AUNCEL_AMD_NO_BYTES=1 python bench.py --no-cpu --no-legs --no-other --steps 24 --in-flight 1 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('fp32 lone batch: %.3f ms/step' % d['ms_per_step'], [(p['kernel'][:18], round(p['ms'], 3)) for p in r.get('per_launch', [])])"
done
unset AUNCEL_AMD_CXXFLAGS
