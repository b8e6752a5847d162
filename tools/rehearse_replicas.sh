#!/bin/bash
# (round 3: the bare command -- `python bench.py --gpus N` starts its own ranks, no torch.distributed.run)
# Rehearsal of the DEFAULT bench.py multi-rank line (--placement replicas, what the driver runs for N = 2, 4, 8) on ONE GPU:
# 2 and 4 ranks sharing cuda:0 over gloo.  Checks that the N > 1 code path (barriers, max-over-ranks timing, the
# watchdog-guarded sharded_verify sub-record) runs and prints one JSON line; the numbers are not scaling measurements.
set -e -o pipefail
mkdir -p gpurun_out
T="timeout -k 10 420"
for N in 2 4; do
  ASD_BENCH_ONE_DEVICE=1 $T python bench.py --gpus $N --dist-backend gloo --steps 20 --warmup 5 \
    > gpurun_out/r03_replicas_n${N}_gloo.json 2> gpurun_out/r03_replicas_n${N}_gloo.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r03_replicas_n${N}_gloo.json").read().strip().splitlines()[-1])
sv=d.get("sharded_verify")
print("replicas N=$N", d["n_gpus"], round(d["value"]), d["unit"], round(d["ms_per_step"]*1e3,2), "us/step", d["scaling"], "sharded_verify:", (sv if not isinstance(sv,dict) else {k: sv[k] for k in list(sv)[:6]}))
PY
done
