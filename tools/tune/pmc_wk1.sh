# Counters of conv1d_wgrad_k1.hip at the train step's three shapes: HBM bytes per launch (FETCH_SIZE / WRITE_SIZE, separate passes, the
# guide's gfx950 corrections) against the algorithmic 8 bytes per element pair, matrix-pipe busy.   usage: bash tools/tune/pmc_wk1.sh <tag>
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-pmc_wk1}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  n=$(echo $c | cut -d' ' -f1)
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/p_$n -o p -- python $R/tools/tune/wk1_probe.py > $O/$n.log 2>&1
  cp $(find $O/p_$n -name "*counter_collection.csv" | head -1) $O/$n.csv
  [ "$n" = FETCH_SIZE ] && cp $(find $O/p_$n -name "*kernel_trace.csv" | head -1) $O/trace.csv
  rm -rf $O/p_$n
done
python - <<PY
import csv, collections, json
def per_dispatch(path, name):
    out = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == name and "wgrad_k1_kernel" in r["Kernel_Name"]:
            out[r["Dispatch_Id"]] = out.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
    return list(out.values())
f, w = per_dispatch("$O/FETCH_SIZE.csv", "FETCH_SIZE"), per_dispatch("$O/WRITE_SIZE.csv", "WRITE_SIZE")
m, g = per_dispatch("$O/SQ_VALU_MFMA_BUSY_CYCLES.csv", "SQ_VALU_MFMA_BUSY_CYCLES"), per_dispatch("$O/SQ_VALU_MFMA_BUSY_CYCLES.csv", "GRBM_GUI_ACTIVE")
dur = [ (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open("$O/trace.csv")) if "wgrad_k1_kernel" in r["Kernel_Name"]]
shapes = [(64, 48000), (96, 48000), (192, 24000)]
res = []
for i, (C, T) in enumerate(shapes):
    sl = slice(4 * i + 1, 4 * i + 4)                       # launches 2 - 4 of each shape
    alg = 2 * 16 * C * T * 4
    fetch = sum(f[sl]) / 3; write = sum(w[sl]) / 3
    hbm = (2 * fetch + write) * 1024                        # MI355X_MICROARCH.md: FETCH_SIZE counts 64 B per 128-B request, KiB units
    busy = (sum(m[sl]) / 3) / ((sum(g[sl]) / 3) / 8 * 1024) if g else None
    res.append({"C": C, "T": T, "B": 16, "algorithmic_bytes": alg, "counter_hbm_bytes": round(hbm), "ratio": round(hbm / alg, 3),
                "kernel_us_under_profiler": round(sum(dur[sl]) / 3, 1), "mfma_busy_of_gui_active": round(busy, 3) if busy else None})
    print(res[-1])
json.dump(res, open("$O/pmc_wk1.json", "w"), indent=1)
PY
rm -f $O/*.csv
