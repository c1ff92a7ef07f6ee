# GPU idle time of the train step (all chains on, as bench.py runs it): kernel trace of 1 warm-up + 3 steps, the last ~2.5 steps analysed
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-idle}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/kt -o t -- python $R/tools/train_bench.py --batch 16 --steps 3 --warmup 2 --predictors > $O/train_line.json 2>$O/kt.err
f=$(find $O/kt -name "*kernel_trace.csv" | head -1)
python $R/tools/gpu_idle.py $f --tail-ms 700 --out $O/gpu_idle_train.json | tail -40
rm -rf $O/kt
