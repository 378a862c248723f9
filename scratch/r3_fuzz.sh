#!/bin/bash
for off in 1000 2000 3000; do
  AUNCEL_TEST_SEED_OFFSET=$off timeout 900 python -m pytest tests/test_gpu_random.py tests/test_gpu_random_adaptive.py -x -q 2>&1 | tail -2
done
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
