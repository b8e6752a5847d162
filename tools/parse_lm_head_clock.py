#!/usr/bin/env python3
"""Join the rocprofv3 passes of tools/r03_lm_head_clock.sh: per lab variant (dispatch order: the libraries are run in sorted
name order -- base, noload, nomath -- the same number of calls each) the mean duration of k_lm_head_tile and its effective
clock  GRBM_GUI_ACTIVE / duration  (and, as a cross-check, SQ_WAVE_CYCLES * 4 / (waves * duration))."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

O = sys.argv[1]
VARIANTS = ["base", "noload", "nomath"]


def rows(path):
    with open(path) as f:
        return list(csv.DictReader(f))


out = {"variants": {}, "how": __doc__.strip()}
per_pass = {}
for pdir in sorted(glob.glob(os.path.join(O, "p*"))):
    if not os.path.isdir(pdir):
        continue
    cc = glob.glob(os.path.join(pdir, "**", "*counter_collection.csv"), recursive=True)
    kt = glob.glob(os.path.join(pdir, "**", "*kernel_trace.csv"), recursive=True)
    if not cc or not kt:
        continue
    times = {}
    for r in rows(kt[0]):
        times[r["Dispatch_Id"]] = (int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"])
    disp = defaultdict(dict)
    for r in rows(cc[0]):
        disp[r["Dispatch_Id"]][r["Counter_Name"]] = disp[r["Dispatch_Id"]].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    seq = []
    for did in sorted(disp, key=lambda d: int(d)):
        if did not in times:
            continue
        st, en, name = times[did]
        if "k_lm_head_tile" not in name:
            continue
        seq.append(dict(ns=en - st, name=name, **disp[did]))
    n = len(seq) // len(VARIANTS)
    for vi, v in enumerate(VARIANTS):
        part = seq[vi * n:(vi + 1) * n]
        part = part[len(part) // 4:]                       # drop the first quarter (warm-up calls)
        # the wide launch (256-column blocks: the long one) is the kernel in question
        wide = [d for d in part if d["ns"] > 0.5 * max(x["ns"] for x in part)]
        agg = {"launches": len(wide), "ns_mean": sum(d["ns"] for d in wide) / max(1, len(wide))}
        for c in wide[0].keys() - {"ns", "name"} if wide else []:
            agg[c] = sum(d.get(c, 0.0) for d in wide) / len(wide)
        per_pass.setdefault(v, {}).update({k: val for k, val in agg.items() if k not in ("launches", "ns_mean")})
        per_pass[v].setdefault("ns_mean_by_pass", []).append(agg["ns_mean"])
        per_pass[v]["launches"] = agg["launches"]
for v, d in per_pass.items():
    ns = sum(d["ns_mean_by_pass"]) / len(d["ns_mean_by_pass"])
    rec = {"wide_launch_us": ns / 1e3, "launches_per_pass": d["launches"], "counters": {k: val for k, val in d.items() if k.isupper()}}
    if "GRBM_GUI_ACTIVE" in d:
        # the counter is reported summed over the chip's 8 XCDs (each XCD's GRBM counts its own shader-clock cycles)
        rec["gui_active_cycles_per_ns_all_xcds"] = d["GRBM_GUI_ACTIVE"] / ns
        rec["clock_GHz"] = d["GRBM_GUI_ACTIVE"] / ns / 8.0
    if "SQ_WAVE_CYCLES" in d and "SQ_WAVES" in d and d["SQ_WAVES"]:
        rec["wave_cycles_per_wave_x4"] = 4.0 * d["SQ_WAVE_CYCLES"] / d["SQ_WAVES"]
    if "SQ_BUSY_CYCLES" in d:
        rec["sq_busy_cycles"] = d["SQ_BUSY_CYCLES"]
    out["variants"][v] = rec
try:
    with open(os.path.join(O, "lab_unprofiled.json")) as f:
        out["unprofiled_call_us"] = {r["variant"]: r["us"] for r in json.load(f)}
except OSError:
    pass
v = out["variants"]
if all(k in v and "clock_GHz" in v[k] for k in VARIANTS):
    c = {k: v[k]["clock_GHz"] for k in VARIANTS}
    out["reading"] = {"clock_GHz": c, "combined_vs_loads_only": c["base"] / c["nomath"], "combined_vs_math_only": c["base"] / c["noload"],
                      "rule": "VERDICT r2 item 7: the combined kernel's clock >= 10 % below BOTH singles => the chip's power management, not "
                              "the schedule, is the limiter at M = 256"}
print(json.dumps(out, indent=1))
with open(os.path.join(O, "summary.json"), "w") as f:
    json.dump(out, f, indent=1)
