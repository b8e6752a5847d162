#!/bin/bash
# X3: HBM traffic of one 7B model pass on the HIP decoder stack (separate --pmc FETCH_SIZE / WRITE_SIZE passes, as
# MI355X_MICROARCH.md prescribes) against the bytes a pass cannot avoid: the weights of every projection once + the lm_head.
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
ARGS="--model 7b --batch 32 --tokens 1 --passes 8 --hip-layers --pack"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/r03_pass_fetch -o p -- python3 $R/tools/profile_pass.py $ARGS > /dev/null 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/r03_pass_write -o p -- python3 $R/tools/profile_pass.py $ARGS > /dev/null 2>&1
cd $R
python3 - <<PY
import csv, glob, json, collections
def load(d, counter):
    path = glob.glob(f"$O/{d}/**/*counter_collection.csv", recursive=True)[0]
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            per[r["Kernel_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"]), int(r["Grid_Size"]) if "Grid_Size" in r else 0))
    return per
f = load("r03_pass_fetch", "FETCH_SIZE"); w = load("r03_pass_write", "WRITE_SIZE")
def short(n): return n.replace("void ", "").replace("asd::(anonymous namespace)::", "").replace("asd::", "").split("(")[0]
out = {"what": "one 7B-shape model pass (B = 32, T = 1) on the HIP decoder stack; the LAST 8 of 11 single-token passes; FETCH_SIZE KiB x1024 x2 (gfx950 half-count of wide streaming reads), WRITE_SIZE KiB x1024; separate --pmc passes", "kernels": {}}
tot_r = tot_w = 0.0
for name, rows in f.items():
    if "asd::" not in name: continue
    rows.sort()
    # the passes after the prefill: the trailing calls (M = 32 forms); take the last 8 passes' worth
    per_pass = {"k_lm_head_skinny": 113, "k_rmsnorm<true>": 56, "k_rmsnorm<false>": 1, "k_rope_kv_store": 28, "k_attn_ragged": 28, "k_silu_mul": 28}
    key = next((k for k in per_pass if k in name), None)
    if key is None: continue
    n = per_pass[key] * 8
    rb = sum(v for _, v, _ in rows[-n:]) * 1024 * 2 / 8
    wrows = sorted(w.get(name, []))
    wb = sum(v for _, v, _ in wrows[-n:]) * 1024 / 8
    out["kernels"][short(name)] = {"launches_per_pass": per_pass[key], "read_GB_per_pass": round(rb / 1e9, 3), "write_GB_per_pass": round(wb / 1e9, 3)}
    tot_r += rb; tot_w += wb
alg = 14.141238272e9 + 152064 * 3584 * 2
out["read_GB_per_pass"] = round(tot_r / 1e9, 3); out["write_GB_per_pass"] = round(tot_w / 1e9, 3)
out["algorithmic_GB_per_pass"] = round(alg / 1e9, 3); out["traffic_over_algorithmic"] = round((tot_r + tot_w) / alg, 4)
json.dump(out, open("$O/r03_pass7b_traffic.json", "w"), indent=1); print(json.dumps(out, indent=1))
PY
