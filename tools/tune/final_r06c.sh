# Final-tree measurement set of round 6, second session (one box, one call): seeded train record, the default bench line, rocprof kernel
# stats of the forward bench and of the train step (every chain serial), LDS bank-conflict counters of the train step.
# usage: bash tools/tune/final_r06c.sh <tag>
TAG=${1:-r06fin4}
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
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/kf -o f -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-train --no-streaming > $O/bench_under_rocprof.json 2>$O/kf.err)
cp $(find $O/kf -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv; rm -rf $O/kf
head -12 $O/bench_kernel_stats.csv | cut -c1-150
bash tools/tune/train_stats.sh ${TAG}_ts > $O/train_stats.log 2>&1; tail -36 $O/train_stats.log | cut -c1-150
(cd /tmp && export TMPDIR=/tmp && FAC_DISC_STREAMS=1 FAC_PRED_STREAMS=1 FAC_QUANT_STREAMS=1 FAC_MEL_STREAMS=1 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O/pl -o l -- python $R/tools/train_bench.py --batch 16 --steps 1 --warmup 1 --predictors > $O/pmc_lds.log 2>$O/pl.err)
python - <<PY
import csv, glob, collections
f = glob.glob("$O/pl/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for fn in f:
    for r in csv.DictReader(open(fn)):
        acc[r["Kernel_Name"].split("(")[0][:60]][r["Counter_Name"]] += float(r["Counter_Value"])
out = {k: {"SQ_LDS_BANK_CONFLICT": v.get("SQ_LDS_BANK_CONFLICT", 0), "SQ_LDS_IDX_ACTIVE": v.get("SQ_LDS_IDX_ACTIVE", 0),
           "conflict_frac": round(v.get("SQ_LDS_BANK_CONFLICT", 0) / v["SQ_LDS_IDX_ACTIVE"], 4) if v.get("SQ_LDS_IDX_ACTIVE") else None}
       for k, v in acc.items() if v.get("SQ_LDS_IDX_ACTIVE", 0) > 1e6}
import json
json.dump(out, open("$O/pmc_lds_conflicts.json", "w"), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -kv[1]["SQ_LDS_IDX_ACTIVE"])[:14]: print(k, v)
PY
rm -rf $O/pl
