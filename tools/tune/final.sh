# Round-end measurement set (run on the GPU box from the repo root): GPU test suite, bench line, kernel stats of the bench and of
# the training step (rocprofv3 kernel trace), FETCH_SIZE / WRITE_SIZE passes for the traffic table, matrix-pipe counters.
# usage: bash tools/tune/final.sh <round tag, e.g. r04> <git sha of the tree>
TAG=${1:-rXX}; SHA=${2:-unknown}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
python bench.py > $O/bench_line.json 2> $O/bench.err; tail -c 300 $O/bench_line.json; echo
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kb -o k -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-train --no-streaming > $O/bench_under_rocprof.json 2>/dev/null
cp $(find $O/kb -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv
FAC_DISC_STREAMS=1 FAC_PRED_STREAMS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o t -- python $R/tools/train_bench.py --batch 16 --steps 3 --warmup 1 --predictors > $O/train_under_rocprof_one_stream.json 2>/dev/null
cp $(find $O/kt -name "*kernel_stats.csv" | head -1) $O/train_kernel_stats_one_stream.csv
CMD="bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-train --no-streaming"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f -o f -- python $R/$CMD > $O/pmc_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_w -o w -- python $R/$CMD > $O/pmc_w.log 2>&1
python $R/tools/pmc_traffic.py $(find $O/pmc_f -name "*counter_collection.csv" | head -1) $(find $O/pmc_w -name "*counter_collection.csv" | head -1) $O/pmc_traffic.json $SHA "$CMD" | tail -12
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/pmc_m -o m -- python $R/$CMD > $O/pmc_m.log 2>&1
python $R/tools/pmc_mfma.py $(find $O/pmc_m -name "*counter_collection.csv" | head -1) $O/pmc_mfma_fwd.json bsplit gemm_split conv1d_pw lstm > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/pmc_l -o l -- python $R/$CMD > $O/pmc_l.log 2>&1
python $R/tools/pmc_mfma.py $(find $O/pmc_l -name "*counter_collection.csv" | head -1) $O/pmc_lds_fwd.json bsplit gemm_split conv1d_pw lstm > /dev/null 2>&1
python $R/tests/tools/cpu_thread_sweep.py > $O/cpu_thread_sweep.log 2>&1
rm -rf $O/pmc_f $O/pmc_w $O/pmc_m $O/pmc_l $O/kb $O/kt
ls $O
echo done
