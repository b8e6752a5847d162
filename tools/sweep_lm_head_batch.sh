#!/bin/bash
# asd_lm_head_verify against the batch size (K = 8; 7B and 72B heads): python tools/bench_lm_head.py per B, one line each.
#   BATCHES="32 33 36 40" bash tools/sweep_lm_head_batch.sh
for B in ${BATCHES:-4 8 12 16 20 24 28 32 33 36 40 48 64}; do
  timeout -k 10 120 python tools/bench_lm_head.py --batch $B --shapes 7b,72b --reps 10 --out gpurun_out/lmh_$B.json > gpurun_out/lmh_$B.log 2>&1 || break
  python - <<PY
import json
rows=[json.loads(l) for l in open("gpurun_out/lmh_$B.log") if l.startswith("{")]
print($B, $B*8, " ".join("%s fused %.0f packed %.0f gemm %.0f" % (r["shape"], r["fused_us"], r.get("fused_packed_us",0), r["gemm_only_us"]) for r in rows))
PY
done
