# is the B = 16 linearity test's quantizer-gradient mismatch a race?  the same test under switches, twice each
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for v in "" "FAC_SYNC_H2D=1" "FAC_SLOW_STREAM=1" "FAC_QUANT_STREAMS=1" "FAC_FOLD_IN_PLACE=0" "FAC_SLOW_STREAM=1 FAC_SYNC_H2D=1" "AMD_SERIALIZE_KERNEL=3"; do
  for rep in 1 2; do
    r=$(env $v python -m pytest tests/test_train_golden.py -m gpu -x -q -k linearity 2>&1 | grep -E "passed|failed|AssertionError: " | tr '\n' ' ')
    echo "[$v] $r"
  done
done
