# one-launch split-reduction conv: parity (conv cases, streaming vs offline, quantizer Linears), streaming latency with / without
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4_skinny; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_cabi.py -x -q -k "conv1d_against or streaming or skinny or rvq or quantizer or end_to_end or cabi or redecoder or predictors or style" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for i in 1 2; do
python - <<PY
import json,sys,torch
sys.path.insert(0,'.')
import bench
m=bench.build(torch.device('cuda:0'))
r=bench.streaming_leg(m, torch.device('cuda:0'), 2000); print('one launch ', r['p50_ms'], r['p99_ms'], r['rtf'])
PY
FAC_SKINNY_ONE_LAUNCH=0 python - <<PY
import json,sys,torch
sys.path.insert(0,'.')
import bench
m=bench.build(torch.device('cuda:0'))
r=bench.streaming_leg(m, torch.device('cuda:0'), 2000); print('two launches', r['p50_ms'], r['p99_ms'], r['rtf'])
PY
done 2>/dev/null
echo done
