# kernel stats of the train step, every chain serial (kernel times add up)   usage: bash tools/tune/train_stats.sh <tag>
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-tstats}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
FAC_DISC_STREAMS=1 FAC_PRED_STREAMS=1 FAC_QUANT_STREAMS=1 FAC_MEL_STREAMS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o t -- python $R/tools/train_bench.py --batch 16 --steps 3 --warmup 1 --predictors > $O/line.json 2>$O/kt.err
cp $(find $O/kt -name "*kernel_stats.csv" | head -1) $O/train_kernel_stats_one_stream.csv
rm -rf $O/kt
python - <<PY
import csv
rows=list(csv.DictReader(open("$O/train_kernel_stats_one_stream.csv")))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('kernel ms per step (4 steps traced):', round(tot/1e6/4,1))
for r in rows[:34]:
    print('%-84s %5d %7.2f %7.1f'%(r['Name'][:84], int(r['Calls'])//4, float(r['TotalDurationNs'])/1e6/4, float(r['AverageNs'])/1e3))
PY
