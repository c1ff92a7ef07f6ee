#!/usr/bin/env python
"""Merges one rocprofv3 kernel trace (durations, un-profiled clocks) with any number of `--pmc` passes of the SAME command into one
per-kernel table: launches, average microseconds, every counter per launch, and the derived figures the design text quotes:

  mfma_busy_pct       = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs)     (rocprofv3's MfmaUtil expression)
  wait_inst_any_frac  = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES   (issue stalls: MFMA dependency / pipe),  wait_any_frac (s_waitcnt / barrier
                        parking), active_inst_any_frac -- the three are disjoint and sum to ~1 (MI355X_MICROARCH.md, PMC slots)
  lds_bank_conflict_frac = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
  hbm_bytes           = (2 * FETCH_SIZE + WRITE_SIZE) * 1024    (gfx950 correction of MI355X_MICROARCH.md, HBM section; both in KiB)
  hbm_TBps            = hbm_bytes / average duration of the un-profiled trace

usage: pmc_kernels.py <out.json> <kernel_stats.csv | kernel_trace.csv> <counter_collection.csv> [more counter csvs ...] [--min-us 0] [--filter name ...]
"""
import collections
import csv
import json
import re
import sys

SIMDS = 256 * 4


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name.replace("fac::", "")


def main():
    argv = sys.argv[1:]
    filt = []
    if "--filter" in argv:
        i = argv.index("--filter")
        filt = argv[i + 1:]
        argv = argv[:i]
    out, trace, passes = argv[0], argv[1], argv[2:]
    dur = {}
    rows = list(csv.DictReader(open(trace)))
    if rows and "AverageNs" in rows[0]:
        for r in rows:
            dur[short(r["Name"])] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3)
    else:
        acc = collections.defaultdict(lambda: [0, 0.0])
        for r in rows:
            a = acc[short(r["Kernel_Name"])]
            a[0] += 1
            a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        dur = {k: (n, t / n) for k, (n, t) in acc.items()}
    ctr = collections.defaultdict(lambda: collections.defaultdict(float))
    nd = collections.defaultdict(lambda: collections.defaultdict(set))
    for p in passes:
        for r in csv.DictReader(open(p)):
            k = short(r["Kernel_Name"])
            ctr[k][r["Counter_Name"]] += float(r["Counter_Value"])
            nd[k][r["Counter_Name"]].add(r["Dispatch_Id"])
    res = {}
    for k, (calls, us) in sorted(dur.items(), key=lambda kv: -kv[1][0] * kv[1][1]):
        if filt and not any(f in k for f in filt):
            continue
        row = {"launches": calls, "avg_us": round(us, 2), "total_ms": round(calls * us / 1e3, 2)}
        c = {name: v / max(1, len(nd[k][name])) for name, v in ctr.get(k, {}).items()}
        if not c:
            res[k] = row
            continue
        row["counters_per_launch"] = {a: round(b, 1) for a, b in sorted(c.items())}
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c and c.get("GRBM_GUI_ACTIVE"):
            row["mfma_busy_pct"] = round(100.0 * c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8.0 * SIMDS), 2)
        if c.get("SQ_WAVE_CYCLES"):
            for nm in ("SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS",
                       "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_MISC", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_FLAT", "SQ_INST_CYCLES_VMEM"):
                if nm in c:
                    row[nm[3:].lower() + "_frac"] = round(c[nm] / c["SQ_WAVE_CYCLES"], 4)
        if c.get("SQ_LDS_IDX_ACTIVE"):
            row["lds_bank_conflict_frac"] = round(c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"], 4)
        if "FETCH_SIZE" in c or "WRITE_SIZE" in c:
            b = (2.0 * c.get("FETCH_SIZE", 0.0) + c.get("WRITE_SIZE", 0.0)) * 1024.0
            row["hbm_bytes"] = round(b)
            row["hbm_TBps"] = round(b / us / 1e6, 3)
            row["hbm_frac_of_8TBps"] = round(b / us / 1e6 / 8.0, 3)
        res[k] = row
    json.dump(res, open(out, "w"), indent=1)
    for k, row in list(res.items())[:45]:
        print("%-64s %5d x %8.1f us  mfma %5s%%  wait_inst %6s  wait %6s  hbm %6s TB/s" % (
            k[:64], row["launches"], row["avg_us"], row.get("mfma_busy_pct", "-"), row.get("wait_inst_any_frac", "-"),
            row.get("wait_any_frac", "-"), row.get("hbm_TBps", "-")))


if __name__ == "__main__":
    main()
