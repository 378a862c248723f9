#!/bin/bash
# Everything the round's committed numbers come from, in one gpurun call:
#   gpurun --timeout 3000 -- 'bash scripts/refresh_round.sh r06'
tag=${1:-r06}
bash profiles/collect.sh $tag > gpurun_out/collect_$tag.log 2>&1
bash profiles/collect_cfg.sh $tag > gpurun_out/collect_cfg_$tag.log 2>&1
cp gpurun_out/summary_$tag/${tag}_pmc_cfg5.json gpurun_out/summary_$tag/${tag}_pmc_cfg3.json profiles/ 2>/dev/null
# (bench.py takes roofline.traffic from the newest profiles/r*_pmc.json: the one this run just measured)
cp gpurun_out/summary_$tag/${tag}_pmc.json profiles/ 2>/dev/null
python bench.py > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.log
python scripts/bench_configs.py > gpurun_out/configs_$tag.jsonl 2> gpurun_out/configs_$tag.log
python bench.py --mode shards --test 10000 > gpurun_out/shards_$tag.json 2> gpurun_out/shards_$tag.log
python scripts/effect_time.py > gpurun_out/effect_time_$tag.jsonl 2> gpurun_out/effect_time_$tag.log
python scratch/latency1.py > gpurun_out/latency1_$tag.txt 2> gpurun_out/latency1_$tag.log
bash scripts/timelines.sh $tag > gpurun_out/timelines_$tag.log 2>&1
bash scripts/busy.sh > gpurun_out/in_flight_busy_$tag.txt 2>&1
bash scripts/chain.sh > gpurun_out/chain_$tag.log 2>&1; mv gpurun_out/r06_chain.txt gpurun_out/chain_$tag.txt 2>/dev/null
bash scripts/repeat_bench.sh $tag 8 > gpurun_out/repeat_$tag.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -q -rf 2>&1 | tail -15 > gpurun_out/gpu_tests_$tag.txt
tail -3 gpurun_out/bench_$tag.log; cat gpurun_out/repeat_$tag.txt | tail -3; du -sh gpurun_out
