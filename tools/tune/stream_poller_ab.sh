# configs[4], 30 minutes of audio, twice: alone, and next to a loop that polls the device through rocm-smi every 5 s -- do the rare
# ~20 ms device-side hops follow a monitoring agent's polling?   -> gpurun_out/<tag>/streaming_poller_ab.json
R=${GRAFT_REPO_ROOT:-/root/repo}
export O=$R/gpurun_out/${1:-poller}; mkdir -p $O
cd $R
python tools/stream_bench.py --minutes 30 > $O/alone.json 2>$O/alone.err
( while true; do rocm-smi --showuse --showmemuse --showtemp --showpower > /dev/null 2>&1; sleep 5; done ) &
POLL=$!
python tools/stream_bench.py --minutes 30 > $O/polled.json 2>$O/polled.err
kill $POLL 2>/dev/null
( while true; do rocm-smi --showuse --showmemuse --showtemp --showpower > /dev/null 2>&1; sleep 0.5; done ) &
POLL=$!
python tools/stream_bench.py --minutes 10 > $O/polled_fast.json 2>$O/polled_fast.err
kill $POLL 2>/dev/null
python - <<'PY'
import json, os
out = {}
for k in ("alone", "polled", "polled_fast"):
    try:
        d = json.loads(open(os.environ["O"] + "/%s.json" % k).read().strip().splitlines()[-1])
        out[k] = {x: d[x] for x in ("hops", "p50_ms", "p99_ms", "p99.9_ms", "max_ms", "slowest_hops_index_host_ms_device_ms", "device_ms_p50", "rtf")}
    except Exception as e:
        out[k] = {"error": str(e)}
out["what"] = "tools/stream_bench.py: alone (30 min of audio) | next to rocm-smi --showuse --showmemuse --showtemp --showpower every 5 s (30 min) | the same poll every 0.5 s (10 min)"
json.dump(out, open(os.environ["O"] + "/streaming_poller_ab.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
