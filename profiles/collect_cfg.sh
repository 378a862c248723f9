#!/bin/bash
# Counters of the float-data kernels (BASELINE configs 5 and 3 at nprobe 32: scan_filter_wide_kernel / scan_filter_kernel, the dense
# scan_tiles_kernel shapes, rescore_kernel, coarse_gemm16_kernel, coarse_pick_kernel), one rocprofv3 pass per counter group:
#   gpurun --timeout 2400 -- 'bash profiles/collect_cfg.sh r05'
# scripts/bench_configs.py searches the batch four times (one warm-up + three timed); profiles/summarize_cfg.py takes every kernel's
# dispatches of the LAST search.  Raw CSVs stay on the box; the summaries go to gpurun_out/summary_<tag>/.
tag=${1:-r05}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/summary_$tag
for cfg in 5 3; do
  out=/tmp/prof_cfg${cfg}_$tag
  rm -rf $out && mkdir -p $out
  run() { # name, counters...
    name=$1; shift
    rocprofv3 "$@" --output-format csv -d $out/$name -- python3 scripts/bench_configs.py --cfg $cfg --nprobes 32 --ref-sample 0 --sample 8 > $out/$name.log 2>&1
  }
  run trace --kernel-trace --stats
  run fetch --pmc FETCH_SIZE
  run write --pmc WRITE_SIZE
  run sq --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
  run sq2 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INST_CYCLES_VMEM
  run misc --pmc GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SMEM
  run l2 --pmc TCC_HIT_sum TCC_MISS_sum
  python3 profiles/summarize_cfg.py $out $cfg 4 > gpurun_out/summary_$tag/${tag}_pmc_cfg$cfg.json 2> gpurun_out/summary_$tag/${tag}_pmc_cfg$cfg.err
  tail -1 $out/trace.log | cut -c1-300
  rm -rf $out
done
ls -la gpurun_out/summary_$tag
