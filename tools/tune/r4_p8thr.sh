R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for thr in 700 450 300 150 700; do
FAC_P8_PREPASS=$thr timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-train --no-streaming --no-roofline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('threshold $thr:', d['value'], d['ms_per_step'], d['codes_match'])"
done
echo done
