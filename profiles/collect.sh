#!/bin/bash
# Profiles of bench.py on the GPU box (run through gpurun from the repo root):
#   gpurun --timeout 1800 -- 'bash profiles/collect.sh r01'
# (one batch at a time: the kernels alone on the chip, the configuration bench.py's top-level roofline is measured in)
# One --kernel-trace run for per-kernel durations, and one run per PMC group (FETCH_SIZE and WRITE_SIZE do
# not fit one pass on gfx950: MI355X_MICROARCH.md "rocprofv3 PMC slots").  Raw CSVs land in gpurun_out/;
# profiles/summarize_trace.py and profiles/summarize_pmc.py turn them into the committed summaries.
set -x
tag=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_$tag
rm -rf $out && mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python bench.py --no-cpu --no-legs --in-flight 1 --steps 5 > $out/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python bench.py --no-cpu --no-legs --in-flight 1 --steps 5 > $out/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- python bench.py --no-cpu --no-legs --in-flight 1 --steps 5 > $out/write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $out/sq -- python bench.py --no-cpu --no-legs --in-flight 1 --steps 5 > $out/sq.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SMEM --output-format csv -d $out/misc -- python bench.py --no-cpu --no-legs --in-flight 1 --steps 5 > $out/misc.log 2>&1
# L2 / L1 view of the scans (hit rate = TCC_HIT / (TCC_HIT + TCC_MISS), MI355X_MICROARCH.md L2); a pass whose counters this
# rocprofv3 does not know simply leaves no CSV
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/l2 -- python bench.py --no-cpu --no-legs --in-flight 1 --steps 5 > $out/l2.log 2>&1
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum --output-format csv -d $out/l1 -- python bench.py --no-cpu --no-legs --in-flight 1 --steps 5 > $out/l1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INST_CYCLES_VMEM --output-format csv -d $out/sq2 -- python bench.py --no-cpu --no-legs --in-flight 1 --steps 5 > $out/sq2.log 2>&1
# gpurun brings back at most 64 MiB: summarise here, keep the summaries, the kernel trace and its stats, drop the raw counter dumps
python profiles/summarize.py $out $tag gpurun_out/summary_$tag > $out/summarize.log 2>&1
cp $out/trace/*/*_kernel_stats.csv gpurun_out/summary_$tag/${tag}_kernel_stats.csv
rm -rf $out/fetch $out/write $out/sq $out/misc $out/l2 $out/l1 $out/sq2
find $out gpurun_out/summary_$tag -type f | head -30
