#!/bin/bash
# copy what scripts/refresh_round.sh brought back under gpurun_out/ into the tracked profiles/ directory
tag=${1:-r06}
cp gpurun_out/summary_$tag/${tag}_summary.md gpurun_out/summary_$tag/${tag}_pmc.json gpurun_out/summary_$tag/${tag}_kernel_stats.csv profiles/
cp gpurun_out/summary_$tag/${tag}_pmc_cfg5.json gpurun_out/summary_$tag/${tag}_pmc_cfg3.json profiles/ 2>/dev/null
cp gpurun_out/bench_$tag.json profiles/${tag}_bench_line.json
cp gpurun_out/configs_$tag.jsonl profiles/${tag}_other_configs.jsonl
cp gpurun_out/shards_$tag.json profiles/${tag}_shards_one_gpu.json
cp gpurun_out/effect_time_$tag.jsonl profiles/${tag}_effect_time.jsonl
cp gpurun_out/latency1_$tag.txt profiles/${tag}_latency_batch1.txt
cp gpurun_out/gpu_tests_$tag.txt profiles/${tag}_gpu_tests.txt
grep -v "rocprofv3\|^W2026\|^E2026" gpurun_out/in_flight_busy_$tag.txt > profiles/${tag}_in_flight_busy.txt
cp gpurun_out/chain_$tag.txt profiles/${tag}_chain.txt
cp gpurun_out/bench_lines_repeat_$tag.jsonl profiles/${tag}_bench_lines_repeat.jsonl
cp gpurun_out/repeat_$tag.txt profiles/${tag}_bench_lines_repeat_summary.txt
for n in fp32_path exact_ties id_ties cfg5; do cp gpurun_out/timeline_${n}_$tag.txt profiles/${tag}_timeline_$n.txt; done
true
