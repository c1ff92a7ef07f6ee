# train leg with the exchange forced on a one-rank RCCL group (collectives run, hooks launch them), against the plain run
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4_bench1f; mkdir -p $O; cd $R
FAC_FORCE_ALLREDUCE=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29551 timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-streaming --no-roofline --train-steps 3 --train-warmup 1 > $O/forced.json 2> $O/forced.err; echo "rc=$?"
python - <<PY
import json
d=json.loads([l for l in open('$O/forced.json').read().splitlines() if l.startswith('{')][-1]); t=d['train_step']
print('forced RCCL one rank: train', t.get('ms_per_step'), t.get('allreduce_ms_standalone'), json.dumps(t.get('allreduce_overlap',{}).get('wait_ms')))
PY
echo done
