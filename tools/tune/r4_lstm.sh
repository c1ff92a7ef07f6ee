# split resident LSTM: tests, layer timings, forward bench with / without it, cpu thread sweep, bench watchdog
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4_lstm; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_lstm_persist.py -x -q > $O/pytest_lstm.log 2>&1; tail -5 $O/pytest_lstm.log
FAC_LSTM_PERSIST_MAX_BATCH=32 timeout 300 python tools/lstm_bench.py > $O/lstm_bench.log 2>&1; cat $O/lstm_bench.log | cut -c1-600
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-train --no-streaming > $O/bench_split.json 2> $O/bench_split.err; tail -c 400 $O/bench_split.err; python -c "
import json; d=json.loads(open('$O/bench_split.json').read().strip().splitlines()[-1]); print('split', d['value'], d['ms_per_step'], d['codes_match'], d['codes_match_timed_batch'])"
FAC_LSTM_PERSIST_SPLIT=0 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-train --no-streaming --no-roofline > $O/bench_nosplit.json 2>/dev/null; python -c "
import json; d=json.loads(open('$O/bench_nosplit.json').read().strip().splitlines()[-1]); print('per-step', d['value'], d['ms_per_step'])"
timeout 600 python -m pytest tests/test_train_golden.py -x -q -k "batch32" > $O/pytest_b32.log 2>&1; tail -3 $O/pytest_b32.log
FAC_TRAIN_LEG_TIMEOUT=2 timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-streaming --no-roofline > $O/bench_watchdog.json 2>/dev/null; echo "watchdog rc=$?"; tail -c 300 $O/bench_watchdog.json; echo
timeout 600 python tests/tools/cpu_thread_sweep.py > $O/cpu_thread_sweep.log 2>&1; cat $O/cpu_thread_sweep.log
echo done
