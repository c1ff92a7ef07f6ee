R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/rec3; mkdir -p $O
cd $R
python bench.py --no-cpu-baseline --no-streaming --record-train-loss > $O/record_line.json 2> $O/record.err
cp tests/golden/bench_train_seeded.json $O/bench_train_seeded.json
cat $O/bench_train_seeded.json
FAC_WEIGHT_BATCH=0 python bench.py --no-cpu-baseline --no-streaming > $O/line_wb0.json 2> $O/wb0.err
python - <<PY
import json
for f in ("record_line.json","line_wb0.json"):
    d=json.loads(open("$O/"+f).read().strip().splitlines()[-1])
    t=d["train_step"]; print(f, d["value"], t["ms_per_step"], t["loss"], t["mel"], t["loss_seeded"])
PY
