export WARM=5 REP=10
python -m pytest tests -m gpu -x -q -k "split_bf16" 2>&1 | tail -2
SPLIT=1 SEL="RU k7" python tools/conv_bench.py 2>&1 | grep -E "d=1|sum"
