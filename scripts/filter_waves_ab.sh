# scan_filter_kernel at 2 / 3 / 4 waves a SIMD (libraries built with -DAUNCEL_FILTER_WAVES=n): fp32_path of bench.py + cfg 3
R=${GRAFT_REPO_ROOT:-$(pwd)}
for rep in 1 2; do
for w in 2 3 4; do
  if [ $w = 3 ]; then unset AUNCEL_AMD_LIB; else export AUNCEL_AMD_LIB=$R/auncel_amd/lib/libauncel_amd_fw$w.so; fi
  AUNCEL_BENCH_FP32_STEPS=24 AUNCEL_BENCH_SKIP_LEGS=one_batch,id_ties,fixed,latency1 python bench.py --steps 6 --warmup 6 --no-cpu --no-other 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); f=j['fp32_path']
print('waves', '$w', 'fp32', round(f['value']), 'ms', round(f['ms_per_step'],3), 'alone', round(f['one_batch_at_a_time_ms'],3), {k: round(v,3) for k,v in f['roofline']['phases_ms_per_step'].items() if 'scan' in k})"
done; done
for w in 2 3 4; do
  if [ $w = 3 ]; then unset AUNCEL_AMD_LIB; else export AUNCEL_AMD_LIB=$R/auncel_amd/lib/libauncel_amd_fw$w.so; fi
  python scripts/bench_configs.py --cfg 3 --nprobes 32 --ref-sample 0 --sample 8 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('waves', '$w', 'cfg3', round(j['value']), {k: round(v['ms'],3) for k,v in j['phases'].items()})"
done
