# bench.py's multi-rank path with two ranks sharing the one GPU over gloo (no RCCL between two ranks of one device): barriers, MAX-over-ranks
# timing, aggregated units, rank-0 line, train leg with the hook-time exchange and its stopwatch
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4_bench2; mkdir -p $O; cd $R
export FAC_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --steps 3 --warmup 1 --train-steps 2 --train-warmup 1 > $O/bench_2ranks_gloo_one_gpu.json 2> $O/bench.err; echo "rc=$?"; tail -5 $O/bench.err | cut -c1-300
python - <<PY
import json
lines=[l for l in open('$O/bench_2ranks_gloo_one_gpu.json').read().splitlines() if l.startswith('{')]
print(len(lines),'json line(s)')
d=json.loads(lines[-1])
print(d['n_gpus'], d['value'], d['ms_per_step'], d['config']['parallelism'])
t=d.get('train_step',{})
print({k:t.get(k) for k in ('value','ms_per_step','allreduce_bytes_per_step','allreduce_ms_standalone','error')})
print(json.dumps(t.get('allreduce_overlap'))[:1500])
PY
echo done
