# forward (configs[1]) A/B on one box, B first then alternating three times: $2 = environment of the B leg
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-fwd_abab}; mkdir -p $O
cd $R
for i in 1 2 3; do
  env $2 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-train --no-streaming 2>>$O/err.log | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('$2', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['conv_ms_per_step'])" | tee -a $O/ab.log
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-train --no-streaming 2>>$O/err.log | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('baseline', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['conv_ms_per_step'])" | tee -a $O/ab.log
done
