# two ranks of the train step on the one GPU (gloo): bucketed hook-time exchange, side-stream join, rank-asymmetric masks
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4_ddp; mkdir -p $O; cd $R
export FAC_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 tools/ddp_smoke.py > $O/ddp_smoke.log 2>&1; echo "rc=$?"; tail -3 $O/ddp_smoke.log | cut -c1-1500
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29518 tools/ddp_smoke.py --predictors > $O/ddp_smoke_predictors.log 2>&1; echo "rc=$?"; tail -3 $O/ddp_smoke_predictors.log | cut -c1-2500
echo done
