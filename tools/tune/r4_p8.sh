# P8 pre-pass policy: parity, forward bench with / without
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4_p8; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_rccl_one_rank.py -x -q -k "p8 or flattened or slstm or conv_transpose or end_to_end or rccl or one_rank" > $O/pytest.log 2>&1; tail -4 $O/pytest.log
for i in 1 2; do
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-train --no-streaming > $O/bench_on_$i.json 2>/dev/null; python -c "
import json; d=json.loads(open('$O/bench_on_$i.json').read().strip().splitlines()[-1]); r=d['roofline']; print('prepass on ', d['value'], d['ms_per_step'], d['codes_match'], d['codes_match_timed_batch']['mismatches'], {k[:24]:v['ms_per_step'] for k,v in r['all_conv_variants'].items() if 'gemm' in k})"
FAC_P8_PREPASS=0 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-train --no-streaming > $O/bench_off_$i.json 2>/dev/null; python -c "
import json; d=json.loads(open('$O/bench_off_$i.json').read().strip().splitlines()[-1]); r=d['roofline']; print('prepass off', d['value'], d['ms_per_step'], {k[:24]:v['ms_per_step'] for k,v in r['all_conv_variants'].items() if 'gemm' in k})"
done
echo done
