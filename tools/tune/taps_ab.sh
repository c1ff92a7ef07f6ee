# first-layer weight gradients on the virtual-row form of conv1d_wgrad_k1.hip: tests, per-shape listing, train A/B (FAC_WGRAD_K1_STREAM=0 turns
# both this and the k = 1 tails off)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-taps}; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_wgrad_split.py tests/test_train_golden.py -x -q -m gpu 2>&1 | tail -4 | tee $O/test.log
python tools/train_wgrads.py 2>/dev/null > $O/wgrads.log; head -1 $O/wgrads.log; grep "^ *[0-9]* *[12]    \(32\|64\) " $O/wgrads.log | head -20
grep " 64    64   48000.* 1  1  1  0\| 96    96   48000.* 1  1  1  0\|192   192   24000.* 1  1  1  0" $O/wgrads.log
for i in 1 2; do
  for v in 0 1; do
    FAC_WGRAD_K1_STREAM=$v python tools/train_bench.py --batch 16 --steps 6 --warmup 3 --predictors 2>>$O/err.log | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('train FAC_WGRAD_K1_STREAM=$v', d.get('ms_per_step'), d.get('loss'))" | tee -a $O/ab.log
  done
done
