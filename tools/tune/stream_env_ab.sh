# streaming hop A/B on one box, alternating: $2 = environment of the B leg; $3 = minutes per run
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-stream_env_ab}; mkdir -p $O
cd $R
for i in 1 2 3; do
  for leg in "X=1" "$2"; do
    env $leg python tools/stream_bench.py --minutes ${3:-1} 2>>$O/err.log | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$leg', {k:d[k] for k in ('p50_ms','p90_ms','p99_ms','max_ms','device_ms_p50','rtf')})" | tee -a $O/ab.log
  done
done
