# k = 7 kernel walking several tiles per workgroup: parity, layer timings against the one-tile build, forward bench
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4_persist; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "split or conv1d_against or residual_unit or p8 or end_to_end or encoder_decoder" > $O/pytest_conv.log 2>&1; tail -4 $O/pytest_conv.log
for i in 1 2; do
python tools/tune/abl_bsplit.py 2>/dev/null | tail -1
FAC_LIB_PATH=$R/facodec_amd/libfacodec_hip_nopersist.so python tools/tune/abl_bsplit.py 2>/dev/null | tail -1
done > $O/abl.log 2>&1; cat $O/abl.log
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-train --no-streaming > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.err; python -c "
import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); r=d['roofline']; print('persist', d['value'], d['ms_per_step'], d['codes_match'], d['codes_match_timed_batch']['mismatches'], r['achieved'], r['frac'], {k[:28]:v['ms_per_step'] for k,v in r['all_conv_variants'].items()})"
FAC_LIB_PATH=$R/facodec_amd/libfacodec_hip_nopersist.so timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-train --no-streaming > $O/bench_np.json 2>/dev/null; python -c "
import json; d=json.loads(open('$O/bench_np.json').read().strip().splitlines()[-1]); r=d['roofline']; print('one tile', d['value'], d['ms_per_step'], r['achieved'], r['frac'])"
timeout 600 python -m pytest tests/test_train_golden.py -x -q > $O/pytest_train.log 2>&1; tail -3 $O/pytest_train.log
echo done
