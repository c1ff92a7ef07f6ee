# batched weight preparation (FAC_WEIGHT_BATCH) A/B on one box: new tests, forward alternating, train step alternating
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-wb_ab}; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_weight_batch.py -x -q -m gpu 2>&1 | tail -15 | tee $O/test.log
for i in 1 2; do
  for wb in 0 1; do
    FAC_WEIGHT_BATCH=$wb python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-train --no-streaming 2>>$O/err.log | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('forward FAC_WEIGHT_BATCH=$wb', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['conv_ms_per_step'])" | tee -a $O/ab.log
  done
done
for i in 1 2; do
  for wb in 0 1; do
    FAC_WEIGHT_BATCH=$wb python tools/train_bench.py --batch 16 --steps 6 --warmup 3 --predictors 2>>$O/err.log | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('train FAC_WEIGHT_BATCH=$wb', d.get('ms_per_step'), d.get('loss'))" | tee -a $O/ab.log
  done
done
tail -5 $O/err.log
