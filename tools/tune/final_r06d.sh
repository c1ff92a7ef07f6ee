# Final-tree measurement set of round 6, third session (one box, one call): the default bench line, rocprof kernel stats of the forward
# bench and of the train step (every chain serial).   usage: bash tools/tune/final_r06d.sh <tag>
TAG=${1:-r06fin5}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
t0=$(date +%s)
python bench.py --detail $O/bench_detail.json > $O/bench_line.json 2> $O/bench.err
echo "default bench wall seconds: $(( $(date +%s) - t0 ))" | tee $O/bench_wall.txt
python - <<PY
import json
d=json.loads(open("$O/bench_line.json").read().strip().splitlines()[-1])
print(len(open("$O/bench_line.json").read()), d["value"], d["ms_per_step"], d["roofline"]["frac"], d.get("train_step",{}).get("ms_per_step"), d.get("codes_match"))
print(json.dumps(d.get("train_step", {}))[:900])
print(json.dumps(d.get("streaming", d.get("configs4", {})))[:600])
PY
tail -3 $O/bench.err
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/kf -o f -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-train --no-streaming > $O/bench_under_rocprof.json 2>$O/kf.err)
cp $(find $O/kf -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv; rm -rf $O/kf
head -14 $O/bench_kernel_stats.csv | cut -c1-150
bash tools/tune/train_stats.sh ${TAG}_ts > $O/train_stats.log 2>&1; tail -36 $O/train_stats.log | cut -c1-150
cp $R/gpurun_out/${TAG}_ts/train_kernel_stats_one_stream.csv $O/ 2>/dev/null
