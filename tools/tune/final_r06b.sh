# Final-tree measurement set of round 6 (one box, one call): seeded train record, the default bench line (its own check of that record
# included), kernel stats of the train step with every chain serial, GPU idle analysis.   usage: bash tools/tune/final_r06b.sh <tag>
TAG=${1:-r06fin3}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
python bench.py --no-cpu-baseline --no-streaming --record-train-loss > $O/record_line.json 2> $O/record.err
cp tests/golden/bench_train_seeded.json $O/bench_train_seeded.json
t0=$(date +%s)
python bench.py --detail $O/bench_detail.json > $O/bench_line.json 2> $O/bench.err
echo "default bench wall seconds: $(( $(date +%s) - t0 ))" | tee $O/bench_wall.txt
python - <<PY
import json
d=json.loads(open("$O/bench_line.json").read().strip().splitlines()[-1])
print(len(open("$O/bench_line.json").read()), d["value"], d["ms_per_step"], d["roofline"]["frac"], d.get("train_step",{}).get("ms_per_step"), d.get("codes_match"))
print(json.dumps(d.get("streaming", d.get("configs4", {})))[:600])
PY
bash tools/tune/train_stats.sh ${TAG}_ts > $O/train_stats.log 2>&1; tail -36 $O/train_stats.log | cut -c1-150
bash tools/tune/r6_idle.sh ${TAG}_idle > $O/idle.log 2>&1; head -12 $O/idle.log
