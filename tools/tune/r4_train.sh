# elementwise passes of the train step without 64-bit divisions: parity, train step timing, kernel stats
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4_train; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_train_golden.py tests/test_wgrad_split.py -x -q > $O/pytest_train.log 2>&1; tail -3 $O/pytest_train.log
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "wgrad or weight_grad or bwd or backward or discriminator" > $O/pytest_bwd.log 2>&1; tail -3 $O/pytest_bwd.log
for i in 1 2; do timeout 300 python tools/train_bench.py --batch 16 --steps 6 --warmup 2 --predictors 2>/dev/null | tail -1 | cut -c1-200; done
cd /tmp && export TMPDIR=/tmp
FAC_DISC_STREAMS=1 FAC_PRED_STREAMS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o t -- python $R/tools/train_bench.py --batch 16 --steps 3 --warmup 1 --predictors > $O/train_under_rocprof.json 2>/dev/null
cp $(find $O/kt -name "*kernel_stats.csv" | head -1) $O/train_kernel_stats_one_stream.csv; rm -rf $O/kt
python - <<PY
import csv
rows=list(csv.DictReader(open('$O/train_kernel_stats_one_stream.csv')))
print(sum(float(r['TotalDurationNs']) for r in rows)/4e6,'ms kernel time per step')
for r in rows:
    if any(k in r['Name'] for k in ('split_planes','pad_fold','zero_insert')): print(r['Name'][:60], int(r['Calls'])//4, round(float(r['TotalDurationNs'])/4e6,2))
PY
echo done
