#!/usr/bin/env python
"""Per-kernel matrix-pipe utilisation from a rocprofv3 --pmc pass (csv output):
    MFMA busy % = sum(SQ_VALU_MFMA_BUSY_CYCLES) / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs)
(rocprofv3's MfmaUtil expression; the csv row of GRBM_GUI_ACTIVE is the SUM over the 8 XCDs -- calibrated against the
dispatch's own timestamps: GRBM_GUI_ACTIVE / duration = 18.7 cycles per ns = 8 x 2.34 GHz), plus the raw SQ counters per dispatch.  usage: pmc_mfma.py <counter_collection.csv> <out.json> [name filter ...]"""
import collections
import csv
import json
import sys

SIMDS = 256 * 4


def main():
    path, out = sys.argv[1], sys.argv[2]
    filt = sys.argv[3:]
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        if filt and not any(f in k for f in filt):
            continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k].add(r["Dispatch_Id"])
    res = {}
    for k, c in acc.items():
        n = len(disp[k])
        row = {"dispatches": n}
        row.update({name: v / n for name, v in sorted(c.items())})
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c:
            row["mfma_busy_pct_of_all_simds"] = round(100.0 * c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8.0 * SIMDS), 2)
        if "SQ_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c:
            row["sq_busy_over_gui_active"] = round(c["SQ_BUSY_CYCLES"] / c["GRBM_GUI_ACTIVE"], 3)
        if "SQ_WAVE_CYCLES" in c:
            for nm in ("SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
                if nm in c:
                    row[nm.lower() + "_frac_of_wave_cycles"] = round(c[nm] / c["SQ_WAVE_CYCLES"], 3)
        if "SQ_LDS_BANK_CONFLICT" in c and c.get("SQ_LDS_IDX_ACTIVE"):
            row["lds_bank_conflict_frac"] = round(c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"], 4)
        res[k] = row
    json.dump(res, open(out, "w"), indent=1)
    for k, row in res.items():
        print(k[:70], {a: b for a, b in row.items() if a in ("dispatches", "mfma_busy_pct_of_all_simds", "sq_busy_over_gui_active", "lds_bank_conflict_frac")})


if __name__ == "__main__":
    main()
