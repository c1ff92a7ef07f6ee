cd /tmp && export TMPDIR=/tmp
R=/root/repo
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/tstats -o t -- python $R/tools/train_bench.py --batch 16 --steps 2 --warmup 1 > $R/gpurun_out/tstats.log 2>&1
tail -1 $R/gpurun_out/tstats.log | cut -c1-400
