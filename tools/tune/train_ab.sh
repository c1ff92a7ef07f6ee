# train-step A/B on one box, alternating: $2 = the environment setting of the B leg (e.g. FAC_NARROW_TWO_LEVEL=0)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-train_ab}; mkdir -p $O
cd $R
for i in 1 2; do
  python tools/train_bench.py --batch 16 --steps 6 --warmup 2 --predictors 2>>$O/err.log | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('baseline', d.get('ms_per_step'))" | tee -a $O/ab.log
  env $2 python tools/train_bench.py --batch 16 --steps 6 --warmup 2 --predictors 2>>$O/err.log | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('$2', d.get('ms_per_step'))" | tee -a $O/ab.log
done
