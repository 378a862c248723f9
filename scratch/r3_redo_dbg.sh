#!/bin/bash
AUNCEL_AMD_DEBUG_REDO=1 AUNCEL_AMD_COARSE_TIES=redo timeout 600 python bench.py --no-cpu --no-legs --steps 12 --warmup 2 --in-flight 1 2>gpurun_out/redo_err.txt | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); c=j['config']
print('ties redo in-flight 1', 'q/s %.0f ms/step %.3f recall %.4f' % (j['value'], j['ms_per_step'], c['recall_at_10_mean']))"
grep "\[redo\]" gpurun_out/redo_err.txt | tail -12
