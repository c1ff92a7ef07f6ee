#!/usr/bin/env python
"""How much of a traced interval the GPU spent with NO kernel running, from a rocprofv3 `--kernel-trace --output-format csv`
kernel_trace.csv (Start_Timestamp / End_Timestamp per dispatch, all streams): the union of the kernels' intervals against the span
of the last `--steps` repetitions of the step (found by the periodic first kernel of a step, or simply the whole trace), the
distribution of the gaps, and the kernels that most often sit BEHIND a long gap (the host, or a dependency on another stream, was
late for them).   python tools/gpu_idle.py kernel_trace.csv [--skip-first-ms 0] [--gap-us 10]"""
import argparse
import collections
import csv
import json
import re


def short(name):
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*$", "", name).replace("fac::", "")[:70]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--tail-ms", type=float, default=0.0, help="only the last N ms of the trace (0 = all): e.g. the timed steps behind the warm-up")
    ap.add_argument("--gap-us", type=float, default=10.0)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    ev = []
    for r in csv.DictReader(open(a.trace)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
    ev.sort()
    t_end = max(e[1] for e in ev)
    if a.tail_ms > 0:
        ev = [e for e in ev if e[0] >= t_end - a.tail_ms * 1e6]
    t0 = ev[0][0]
    span = (t_end - t0) / 1e6
    busy, cur_end, gaps, last_name = 0, ev[0][0], [], ev[0][2]
    sum_dur = 0
    longest = []
    for s, e, n in ev:
        sum_dur += e - s
        if s > cur_end:
            gaps.append((s - cur_end, n))
            longest.append((s - cur_end, round((cur_end - t0) / 1e6, 2), last_name, n))
        if e > cur_end:
            busy += e - max(s, cur_end)
            cur_end = e
            last_name = n
    longest.sort(reverse=True)
    idle = span - busy / 1e6
    big = [g for g in gaps if g[0] >= a.gap_us * 1e3]
    behind = collections.Counter()
    for g, n in big:
        behind[n] += g
    hist = collections.Counter()
    for g, _ in gaps:
        us = g / 1e3
        hist["<2us" if us < 2 else "2-5us" if us < 5 else "5-10us" if us < 10 else "10-50us" if us < 50 else "50-200us" if us < 200 else ">=200us"] += g
    res = {"dispatches": len(ev), "span_ms": round(span, 2), "busy_union_ms": round(busy / 1e6, 2), "idle_ms": round(idle, 2),
           "idle_frac": round(idle / span, 4), "sum_of_kernel_durations_ms": round(sum_dur / 1e6, 2),
           "overlap_ms (kernels running side by side)": round((sum_dur - busy) / 1e6, 2),
           "idle_ms_by_gap_length": {k: round(v / 1e6, 2) for k, v in hist.items()},
           "gaps": len(gaps), f"gaps_over_{a.gap_us:g}us": len(big),
           "idle_ms_in_front_of (top 12)": {k: round(v / 1e6, 2) for k, v in behind.most_common(12)},
           "longest_gaps [us, at ms, kernel that ended last, kernel that started]": [[round(g / 1e3, 1), at, a_[:48], b_[:48]] for g, at, a_, b_ in longest[:24]]}
    print(json.dumps(res, indent=1))
    if a.out:
        json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
