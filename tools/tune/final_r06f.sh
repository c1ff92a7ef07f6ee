# rocprofv3 kernel stats of the final tree: forward bench and train step (every chain serial)   usage: bash tools/tune/final_r06f.sh <tag>
TAG=${1:-r06fin9}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG; mkdir -p $O
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/kf -o f -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-train --no-streaming > $O/bench_under_rocprof.json 2>$O/kf.err)
cp $(find $O/kf -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv; rm -rf $O/kf
head -6 $O/bench_kernel_stats.csv | cut -c1-150
cd $R; bash tools/tune/train_stats.sh ${TAG}_ts > $O/train_stats.log 2>&1; tail -36 $O/train_stats.log | cut -c1-150 | head -24
cp $R/gpurun_out/${TAG}_ts/train_kernel_stats_one_stream.csv $O/ 2>/dev/null
