# FETCH_SIZE of the weight-gradient GEMM kernel with the plain 3-D grid and with the XCD-aware order (tools/wgrad_bench.py shapes).
# usage: bash tools/tune/pmc_wgrad.sh <out dir under gpurun_out>
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-pmc_wgrad}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
  FAC_WGRAD_XCD=$v FAC_WGRAD_KSPLIT=0 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/f$v -o f -- python $R/tools/wgrad_bench.py > $O/run_$v.log 2>&1
  cp $(find $O/f$v -name "*counter_collection.csv" | head -1) $O/fetch_$v.csv
  rm -rf $O/f$v
done
python - <<PY
import csv, collections
for v in (0, 1):
    tot, n = collections.Counter(), collections.Counter()
    for r in csv.DictReader(open("$O/fetch_%d.csv" % v)):
        if r["Counter_Name"] == "FETCH_SIZE" and "kmajor_kernel" in r["Kernel_Name"]:
            tot["kmajor"] += float(r["Counter_Value"]); n["kmajor"] += 1
    print("FAC_WGRAD_XCD=%d: conv1d_wgrad_kmajor_kernel launches %d, FETCH_SIZE sum %.1f GiB-as-counted (KiB units), per launch %.1f MiB" % (v, n["kmajor"], tot["kmajor"] / 2**20, tot["kmajor"] / max(1, n["kmajor"]) / 1024))
PY
rm -f $O/fetch_0.csv $O/fetch_1.csv
