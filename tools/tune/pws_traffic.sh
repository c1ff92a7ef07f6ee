# HBM traffic (counters) of the streaming bf16-plane kernels at the forward's shapes against their algorithmic bytes.
# Separate --pmc passes for FETCH_SIZE and WRITE_SIZE (MI355X_MICROARCH.md, HBM section).   usage: bash tools/tune/pws_traffic.sh <tag>
TAG=${1:-pws_traffic}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/$c -o p -- python $R/tools/tune/pws_ab.py > $O/$c.log 2>$O/$c.err
done
python - <<PY
import csv, glob, json, collections
def load(counter):
    acc = collections.defaultdict(list)
    for fn in glob.glob("$O/%s/**/*counter_collection.csv" % counter, recursive=True):
        for r in csv.DictReader(open(fn)):
            if r["Counter_Name"] != counter: continue
            k = r["Kernel_Name"]
            if "pws_kernel" not in k and "pwt_kernel" not in k: continue
            acc[(k.split("(")[0].replace("void fac::", ""), r.get("LDS_Block_Size", r.get("LDS_Block_Size_v", "?")), r.get("Grid_Size", "?"))].append(float(r["Counter_Value"]))
    return acc
f, w = load("FETCH_SIZE"), load("WRITE_SIZE")
B = 32
alg = {}   # (kernel template, lds bytes) -> algorithmic bytes: filled by hand below from the shapes of pws_ab.py
out = []
for k in sorted(f):
    fe = sum(f[k]) / len(f[k]); wr = sum(w.get(k, [0])) / max(1, len(w.get(k, [0])))
    out.append(dict(kernel=k[0], lds_bytes=k[1], grid=k[2], launches=len(f[k]), fetch_KiB_raw=round(fe, 1), write_KiB_raw=round(wr, 1),
                    hbm_bytes=round((2 * fe + wr) * 1024), correction="2*FETCH_SIZE + WRITE_SIZE, KiB -> bytes"))
json.dump(out, open("$O/pws_traffic.json", "w"), indent=1)
for o in out: print(o)
PY
rm -rf $O/FETCH_SIZE $O/WRITE_SIZE
