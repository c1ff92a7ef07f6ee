export WARM=5 REP=10
for v in BASE NSW8 BASE NSW8; do
  if [ $v = BASE ]; then unset FAC_LIB_PATH; else export FAC_LIB_PATH=/root/repo/abl/lib_$v.so; fi
  echo "== $v"; SPLIT=1 SEL="RU k7" python tools/conv_bench.py 2>&1 | grep -E "C=768 T=960 d=1|C=384 T=4800 d=1|C=256 T=4800 d=1"
done
