# split-reduction conv geometry sweep on the streaming hop (serial chain = clean signal; then the two-chain hop for the best few)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-skinny_sweep}; mkdir -p $O
cd $R
run() {  # $1 = label, rest = env
  env "${@:2}" python tools/stream_bench.py --minutes 0.3 2>>$O/err.log | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$*', {k:d[k] for k in ('p50_ms','p90_ms','device_ms_p50','rtf')})" | tee -a $O/sweep.log
}
for u in 8 16 32; do for mr in 32 64 128; do for wg in 512 256; do
  run serial FAC_STREAM_TWO_STREAMS=0 FAC_SKINNY_U=$u FAC_SKINNY_MIN_ROWS=$mr FAC_SKINNY_WGS=$wg
done; done; done
for u in 8 32; do for mr in 32 64 128; do
  run two FAC_SKINNY_U=$u FAC_SKINNY_MIN_ROWS=$mr
done; done
