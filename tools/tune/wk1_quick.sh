R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-wk1_q}; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_wgrad_split.py -x -q -m gpu 2>&1 | tail -3 | tee $O/test.log
for i in 1 2; do FAC_WGRAD_K1_STREAM=1 timeout 600 python tools/wgrad_bench.py 2>>$O/err.log | grep "k1 T\(24\|48\)" | cut -c1-110 | tee -a $O/wgrad_bench.log; done
