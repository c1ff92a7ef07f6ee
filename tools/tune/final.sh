R=/root/repo
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err; tail -c 600 gpurun_out/bench_final.json; echo
python tools/train_bench.py --batch 16 --steps 3 > gpurun_out/train_bench.json 2>/dev/null; cat gpurun_out/train_bench.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kstats -o k -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline > $R/gpurun_out/kstats.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/tstats -o t -- python $R/tools/train_bench.py --batch 16 --steps 2 --warmup 1 > $R/gpurun_out/tstats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_f -o f -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $R/gpurun_out/pmc_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_w -o w -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $R/gpurun_out/pmc_w.log 2>&1
echo done
