# GPU idle time of the configs[1] forward step (bench.py's own loop, no roofline events): kernel trace of 3 warm-up + 10 steps, the last ~8 analysed
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-idle_fwd}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/kt -o t -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-train --no-streaming --no-roofline > $O/line.json 2>$O/kt.err
f=$(find $O/kt -name "*kernel_trace.csv" | head -1)
python $R/tools/gpu_idle.py $f --tail-ms 440 --out $O/gpu_idle_fwd.json | tail -60
rm -rf $O/kt
