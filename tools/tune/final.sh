# Round-end measurement set (run on the GPU box from the repo root): GPU test suite, bench line, kernel stats of the bench
# and of the training step (rocprofv3 kernel trace -> tools/rocpd_stats.py), FETCH_SIZE / WRITE_SIZE passes for the traffic table.
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python bench.py > $R/gpurun_out/bench_final.json 2> $R/gpurun_out/bench_final.err; tail -c 400 $R/gpurun_out/bench_final.json; echo
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d /tmp/kb -o k -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-train > $R/gpurun_out/bench_under_rocprof.json 2>/dev/null
python $R/tools/rocpd_stats.py $(find /tmp/kb -name "*.db" | head -1) $R/gpurun_out/bench_kernel_stats.csv
rocprofv3 --kernel-trace -d /tmp/kt -o t -- python $R/tools/train_bench.py --batch 16 --steps 3 --warmup 1 > $R/gpurun_out/train_under_rocprof.json 2>/dev/null
python $R/tools/rocpd_stats.py $(find /tmp/kt -name "*.db" | head -1) $R/gpurun_out/train_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_f -o f -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-train > $R/gpurun_out/pmc_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_w -o w -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-train > $R/gpurun_out/pmc_w.log 2>&1
python $R/tools/pmc_traffic.py $(find $R/gpurun_out/pmc_f -name "*counter_collection.csv" | head -1) $(find $R/gpurun_out/pmc_w -name "*counter_collection.csv" | head -1) $R/gpurun_out/pmc_traffic.json | tail -25
rm -rf $R/gpurun_out/pmc_f $R/gpurun_out/pmc_w
head -8 $R/gpurun_out/bench_kernel_stats.csv | cut -c1-140
echo done
