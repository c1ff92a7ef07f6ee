# Bench line + forward kernel stats + FETCH / WRITE traffic passes of the current tree (the subset of final.sh that changes when
# only forward launches changed).  usage: bash tools/tune/final_short.sh <tag> <git sha>
TAG=${1:-rXX}; SHA=${2:-unknown}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
python bench.py > $O/bench_line.json 2> $O/bench.err; tail -c 300 $O/bench_line.json; echo
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kb -o k -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-train --no-streaming > $O/bench_under_rocprof.json 2>/dev/null
cp $(find $O/kb -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv
CMD="bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-train --no-streaming"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f -o f -- python $R/$CMD > $O/pmc_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_w -o w -- python $R/$CMD > $O/pmc_w.log 2>&1
python $R/tools/pmc_traffic.py $(find $O/pmc_f -name "*counter_collection.csv" | head -1) $(find $O/pmc_w -name "*counter_collection.csv" | head -1) $O/pmc_traffic.json $SHA "$CMD" | tail -4
rm -rf $O/pmc_f $O/pmc_w $O/kb
ls $O
echo done
