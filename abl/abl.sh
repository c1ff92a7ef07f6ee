export WARM=5 REP=10
for v in NOEPI NOSTAGE_NOLDS_NOEPI; do
  export FAC_LIB_PATH=/root/repo/abl/lib_$v.so
  echo "== $v"; SPLIT=1 SEL="k7 C=768" python tools/conv_bench.py 2>&1 | grep "d=1"; SPLIT=1 SEL="k7 C=128" python tools/conv_bench.py 2>&1 | grep "d=1"
done
