# the linearity test (B = 16 vs two B = 8 halves) N times per variant: round-5 tree (worktree _old) against this tree under switches
R=${GRAFT_REPO_ROOT:-/root/repo}
run() {  # dir, env..., -> pass/fail counts
  d=$1; shift
  p=0; f=0; vals=""
  for rep in 1 2 3 4 5; do
    r=$(cd $d && env "$@" python -m pytest tests/test_train_golden.py -m gpu -x -q -k linearity 2>&1 | grep -E "passed|failed|AssertionError: \(" | tr '\n' ' ')
    case "$r" in *failed*) f=$((f+1)); vals="$vals $(echo $r | sed -n 's/.*AssertionError: (\([^)]*\)).*/\1/p' | cut -c1-60)";; *) p=$((p+1));; esac
  done
  echo "[$d $*] pass $p fail $f $vals"
}
run $R/_old X=1
run $R X=1
run $R FAC_FOLD_IN_PLACE=0
run $R FAC_FOLD_IN_PLACE=2
run $R FAC_FOLD_IN_PLACE=3
run $R FAC_QUANT_STREAMS=1
