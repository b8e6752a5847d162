#!/bin/bash
# (round 3: the bare command -- `python bench.py --gpus N` starts its own ranks, no torch.distributed.run)
# Rehearsal of bench.py --placement sharded-target on ONE GPU: tiny shapes at 1 / 2 / 4 ranks (gloo, host-staged), then
# the real 7B + 72B pair on one rank, and the default line's side measurements.
set -e -o pipefail
mkdir -p gpurun_out
T="timeout -k 10 420"
$T python bench.py --placement sharded-target --tier-shapes tiny,tiny --steps 6 --warmup 2 > gpurun_out/r03_sharded_tiny_n1.json
for N in 2 4; do
  ASD_BENCH_ONE_DEVICE=1 $T python bench.py --gpus $N --placement sharded-target --tier-shapes tiny,tiny --dist-backend gloo \
    --steps 6 --warmup 2 > gpurun_out/r03_sharded_tiny_n${N}_gloo.json 2> gpurun_out/r03_sharded_tiny_n${N}_gloo.err
done
$T python bench.py --placement sharded-target --tier-shapes 7b,72b --steps 6 --warmup 2 > gpurun_out/r03_sharded_real_n1.json
for f in r03_sharded_tiny_n1 r03_sharded_tiny_n2_gloo r03_sharded_tiny_n4_gloo r03_sharded_real_n1; do
  python - <<PY
import json
d=json.load(open("gpurun_out/$f.json")); l=d["loop"]
print("$f", d["n_gpus"], round(d["value"],1), "tok/s", round(l["ms_per_step"],2), "ms/step", l["batch_total"], l["tokens_per_sequence_step"], int(l["bytes_exchanged_per_step_rank0"]), "B/step", round(l["hot_path_share"],4))
PY
done
