cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=/tmp/chainprof; rm -rf $OUT; mkdir -p $OUT
cd $R
timeout 900 rocprofv3 --kernel-trace -d $OUT -o t -- python3 bench.py --no-cpu --no-legs --steps 48 --warmup 12 > $OUT/run.log 2>&1
tail -1 $OUT/run.log | cut -c1-150
python3 scripts/busy.py $OUT/t_results.db 60 10
python3 scripts/chain.py $OUT/t_results.db 40 10 > gpurun_out/r06_chain.txt 2>&1
head -5 gpurun_out/r06_chain.txt
