# weight-gradient row tiles with clamped duplicate rows: duplicates not staged (FAC_WGRAD_SKIP_DUP) -- tests, per-shape listing of the step, train A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-skipdup}; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_wgrad_split.py -x -q -m gpu 2>&1 | tail -3 | tee $O/test.log
for v in 0 1; do
  echo "FAC_WGRAD_SKIP_DUP=$v" | tee -a $O/wgrads.log
  FAC_WGRAD_SKIP_DUP=$v python tools/train_wgrads.py 2>/dev/null | tee -a $O/wgrads_$v.log | head -1 | tee -a $O/wgrads.log
  grep " 32  *[0-9]* *[0-9]* 27 \| 64    64   48000\| 96    96   48000 .* 7 \|192   192   24000 .* 7 " $O/wgrads_$v.log | head -12 | tee -a $O/wgrads.log
done
for i in 1 2; do
  for v in 0 1; do
    FAC_WGRAD_SKIP_DUP=$v python tools/train_bench.py --batch 16 --steps 6 --warmup 3 --predictors 2>>$O/err.log | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('train FAC_WGRAD_SKIP_DUP=$v', d.get('ms_per_step'), d.get('loss'))" | tee -a $O/ab.log
  done
done
