# kernel trace of a short streaming session (graph replay), cut into hops by tools/stream_timeline.py
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-strace}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $R/tools/stream_bench.py --minutes 0.5 > $O/stream_plain.json 2>$O/plain.err
rocprofv3 --kernel-trace --output-format csv -d $O/kt -o t -- python $R/tools/stream_bench.py --minutes 0.05 > $O/stream_line.json 2>$O/kt.err
f=$(find $O/kt -name "*kernel_trace.csv" | head -1)
head -2 $f > $O/trace_head.csv
python $R/tools/stream_timeline.py $f --out $O/stream_timeline.json > $O/timeline.log 2>&1
tail -5 $O/timeline.log
rm -rf $O/kt
