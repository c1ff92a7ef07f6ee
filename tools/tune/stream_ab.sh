# streaming hop A/B on one box: FAC_STREAM_FOLD=0 (separate elementwise launches) vs 1, alternating; $2 = extra env (e.g. FAC_STREAM_TWO_STREAMS=0)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-stream_ab}; mkdir -p $O
cd $R
for i in 1 2; do
  for f in 0 1; do
    env $2 FAC_STREAM_FOLD=$f python tools/stream_bench.py --minutes ${3:-1} 2>>$O/err.log | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$2 fold=$f', {k:d[k] for k in ('p50_ms','p90_ms','p99_ms','max_ms','device_ms_p50','rtf')})" | tee -a $O/ab.log
  done
done
