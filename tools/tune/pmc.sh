cd /tmp && export TMPDIR=/tmp
R=/root/repo
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_f -o f -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $R/gpurun_out/pmc_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_w -o w -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $R/gpurun_out/pmc_w.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kstats -o k -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline > $R/gpurun_out/kstats.log 2>&1
find $R/gpurun_out/pmc_f $R/gpurun_out/pmc_w $R/gpurun_out/kstats -name "*.csv" | head -20
tail -2 $R/gpurun_out/kstats.log
