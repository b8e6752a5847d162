#!/bin/bash
# SQ / LDS / MFMA counters of the N2 kernels at the 72B head (round 2): k_lm_head_tile (M = 256) and k_lm_head_quad (M = 1024).
# Three rocprofv3 --pmc passes of four counters each per batch size (counters only with --kernel-trace, as the pool requires).
set -e -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_lmh_pmc
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY"
P2="SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS"
P3="SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU"
for B in 32 128; do
  i=0
  for P in "$P1" "$P2" "$P3"; do
    i=$((i+1))
    timeout -k 10 400 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $O/b${B}_p$i -o c -- python3 $R/tools/bench_lm_head.py --batch $B --shapes 72b --reps 3 --out $O/b${B}_p$i.json > /dev/null 2>&1
  done
done
cd $R
python3 - <<'PY'
import csv, glob, json, os, collections
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "r02_lmh_pmc")
out = {}
for B in (32, 128):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{O}/b{B}_p*/*counter_collection.csv") + glob.glob(f"{O}/b{B}_p*/*/*counter_collection.csv"):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            if "k_lm_head" not in k: continue
            name = "quad" if "quad" in k else ("skinny" if "skinny" in k else "tile")
            acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
    out[f"M={B*8}"] = {k: {c: sum(v) / len(v) for c, v in d.items()} | {"launches": len(next(iter(d.values())))} for k, d in acc.items()}
json.dump(out, open(os.path.join(O, "summary.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
PY
