#!/bin/bash
# quick PMC passes of bench.py, one batch at a time: per-kernel averages of the last 5 dispatches
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B="python bench.py --no-cpu --steps 5 --warmup 2 --in-flight 1"
o=gpurun_out/pmcq; rm -rf $o; mkdir -p $o
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $o/sq -- $B > $o/sq.log 2>&1
python profiles/pmc_by_kernel.py $o/sq 5 | grep "scan_mfma\|replay_kernel<true, true, [0-9]*, 100"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $o/fetch -- $B > $o/fetch.log 2>&1
python profiles/pmc_by_kernel.py $o/fetch 5 | grep "scan_mfma\|replay_kernel<true, true, [0-9]*, 100"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $o/write -- $B > $o/write.log 2>&1
python profiles/pmc_by_kernel.py $o/write 5 | grep "scan_mfma\|replay_kernel<true, true, [0-9]*, 100"
rm -rf $o
