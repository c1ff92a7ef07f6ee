#!/usr/bin/env python
"""Where a streaming hop's time goes: from a rocprofv3 `--kernel-trace --output-format csv` kernel_trace.csv of
tools/stream_bench.py, the dispatches are cut into hops (the host synchronises between hops: gaps > 150 us), and for the
steady-state hops (graph replay) the per-hop span, the kernels per queue, the sum of their durations and of the gaps in front
of them are listed per kernel name -- the critical chain is the queue whose (durations + gaps) equals the span.
    python tools/stream_timeline.py kernel_trace.csv [--out profiles/rNN_stream_timeline.json]"""
import argparse
import collections
import csv
import json
import re
import statistics


def short(name):
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*$", "", name).replace("fac::", "")[:60]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--out", default=None)
    ap.add_argument("--hop-gap-us", type=float, default=150.0)
    a = ap.parse_args()
    ev = []
    for r in csv.DictReader(open(a.trace)):
        grid = "x".join(str(int(r[f"Grid_Size_{d}"]) // max(1, int(r[f"Workgroup_Size_{d}"]))) for d in "XYZ")
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Queue_Id", "0"), grid))
    ev.sort()
    # a hop starts with the copy of its 480 samples into the session's static input buffer (StreamingCodec.push), outside the graph
    hops, cur = [], []
    for e in ev:
        if e[2] == "__amd_rocclr_copyBuffer" and cur:
            hops.append(cur)
            cur = []
        cur.append(e)
    hops.append(cur)
    sizes = collections.Counter(len(h) for h in hops)
    steady = [h for h in hops[len(hops) // 2:-1] if 150 <= len(h) <= 400]
    spans = [(max(e[1] for e in h) - h[0][0]) / 1e3 for h in steady]
    per_q = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0, 0.0]))   # queue -> kernel -> [n, dur, gap before]
    q_tot = collections.defaultdict(lambda: [0, 0.0, 0.0])
    for h in steady:
        last_end = {}
        for s, e, n, q, _ in h:
            g = (s - last_end[q]) / 1e3 if q in last_end else 0.0
            last_end[q] = e
            r = per_q[q][n]
            r[0] += 1
            r[1] += (e - s) / 1e3
            r[2] += max(g, 0.0)
            t = q_tot[q]
            t[0] += 1
            t[1] += (e - s) / 1e3
            t[2] += max(g, 0.0)
    nh = max(1, len(steady))
    res = {"hops_found": len(hops), "dispatches_per_hop_histogram": dict(sizes.most_common(8)), "steady_hops_analysed": len(steady),
           "span_us_p50": round(statistics.median(spans), 1) if spans else None,
           "queues": {q: {"launches_per_hop": round(t[0] / nh, 1), "kernel_us_per_hop": round(t[1] / nh, 1),
                          "gap_us_per_hop": round(t[2] / nh, 1)} for q, t in q_tot.items()},
           "per_queue_kernels": {q: [{"kernel": n, "per_hop": round(r[0] / nh, 2), "avg_us": round(r[1] / r[0], 2),
                                      "avg_gap_before_us": round(r[2] / r[0], 2), "us_per_hop": round((r[1] + r[2]) / nh, 1)}
                                     for n, r in sorted(d.items(), key=lambda kv: -(kv[1][1] + kv[1][2]))] for q, d in per_q.items()}}
    if steady:
        h = steady[len(steady) // 2]
        t0, last_end, seq = h[0][0], {}, []
        for s, e, n, q, g in h:
            seq.append([q, n, g, round((s - t0) / 1e3, 1), round((e - s) / 1e3, 1), round((s - last_end[q]) / 1e3, 1) if q in last_end else None])
            last_end[q] = e
        res["one_hop [queue, kernel, workgroups, start_us, dur_us, gap_before_us]"] = seq
    print(json.dumps({k: v for k, v in res.items() if not k.startswith("one_hop")}, indent=1))
    if a.out:
        json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
