# Round-6 end-of-round measurement set (one box, one call): GPU tests, the bench line, kernel stats of the forward and of the train
# step, FETCH / WRITE passes for roofline.traffic, matrix-pipe / LDS counters of the forward, the train step's counter table, GPU
# idle analysis, the HBM-kernel table, the host profile.   usage: bash tools/tune/final_r06.sh <tag> <git sha>
TAG=${1:-r06}; SHA=${2:-unknown}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
python bench.py --detail $O/bench_detail.json > $O/bench_line.json 2> $O/bench.err; tail -c 400 $O/bench_line.json; echo
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kb -o k -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-train --no-streaming > $O/bench_under_rocprof.json 2>/dev/null
cp $(find $O/kb -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv
CMD="bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-train --no-streaming"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f -o f -- python $R/$CMD > $O/pmc_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_w -o w -- python $R/$CMD > $O/pmc_w.log 2>&1
python $R/tools/pmc_traffic.py $(find $O/pmc_f -name "*counter_collection.csv" | head -1) $(find $O/pmc_w -name "*counter_collection.csv" | head -1) $O/pmc_traffic.json $SHA "$CMD" | tail -12
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/pmc_m -o m -- python $R/$CMD > $O/pmc_m.log 2>&1
python $R/tools/pmc_mfma.py $(find $O/pmc_m -name "*counter_collection.csv" | head -1) $O/pmc_mfma_fwd.json bsplit gemm_split conv1d_pw lstm > /dev/null 2>&1
rm -rf $O/pmc_f $O/pmc_w $O/pmc_m $O/kb
cd $R
bash tools/tune/pmc_train.sh ${TAG}_pmc > $O/pmc_train.log 2>&1; tail -20 $O/pmc_train.log | cut -c1-180
cp $R/gpurun_out/${TAG}_pmc/pmc_train.json $O/pmc_train.json; cp $R/gpurun_out/${TAG}_pmc/train_kernel_stats.csv $O/train_kernel_stats_one_stream.csv
bash tools/tune/r6_idle.sh ${TAG}_idle > $O/idle.log 2>&1; cp $R/gpurun_out/${TAG}_idle/gpu_idle_train.json $O/gpu_idle_train.json
python tools/elementwise_bench.py --pmc $O/pmc_train.json --out $O/hbm_kernels.json > $O/elementwise.log 2>&1
python tools/tune/host_profile.py 30 > $O/host_profile.log 2>&1
python tools/fwd_layers.py > $O/fwd_layers.log 2>&1
ls $O
echo done
