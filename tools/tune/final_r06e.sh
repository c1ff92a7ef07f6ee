# last call of round 6: full GPU suite, seeded train record re-taken (the first-layer weight gradients are fp32 products now), default bench
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-r06fin8}; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee $O/gpu_tests.log
python bench.py --no-cpu-baseline --no-streaming --record-train-loss > $O/record_line.json 2> $O/record.err
cp tests/golden/bench_train_seeded.json $O/bench_train_seeded.json; cat $O/bench_train_seeded.json
python bench.py --detail $O/bench_detail.json > $O/bench_line.json 2> $O/bench.err
python - <<PY
import json
d=json.loads(open("$O/bench_line.json").read().strip().splitlines()[-1])
t=d["train_step"]
print(len(open("$O/bench_line.json").read()), d["value"], d["ms_per_step"], d["roofline"]["frac"], t["ms_per_step"], t["loss_seeded"], t["reference_check"]["ok"], d["codes_match"], d["streaming"]["p50_ms"])
PY
