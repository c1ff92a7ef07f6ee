# k = 1 small-channel weight gradients on the fp32 streaming kernel (FAC_WGRAD_K1_STREAM): tests, per-shape bench, train A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-wk1_ab}; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_wgrad_split.py tests/test_weight_batch.py -x -q -m gpu 2>&1 | tail -15 | tee $O/test.log
for v in 0 1; do
  echo "FAC_WGRAD_K1_STREAM=$v" | tee -a $O/wgrad_bench.log
  FAC_WGRAD_K1_STREAM=$v timeout 600 python tools/wgrad_bench.py 2>>$O/err.log | grep "k1" | tee -a $O/wgrad_bench.log
done
for i in 1 2; do
  for v in 0 1; do
    FAC_WGRAD_K1_STREAM=$v python tools/train_bench.py --batch 16 --steps 6 --warmup 3 --predictors 2>>$O/err.log | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('train FAC_WGRAD_K1_STREAM=$v', d.get('ms_per_step'), d.get('loss'))" | tee -a $O/ab.log
  done
done
tail -5 $O/err.log
