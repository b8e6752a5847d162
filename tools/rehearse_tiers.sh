#!/bin/bash
# (round 3: the bare command -- `python bench.py --gpus N` starts its own ranks, no torch.distributed.run)
# Rehearsal of bench.py --placement tiers on ONE GPU: 1 rank (tiny, then the real 7B/32B/72B shapes) and 2 / 4 ranks
# sharing cuda:0 over gloo (host-staged messages) -- the HIP kernels + the multi-rank protocol in one run.
set -e -o pipefail
mkdir -p gpurun_out
T="timeout -k 10 420"
$T python bench.py --placement tiers --tier-shapes tiny,tiny,tiny --steps 6 --warmup 2 > gpurun_out/r03_tiers_tiny_n1.json
for N in 2 4; do
  ASD_BENCH_ONE_DEVICE=1 $T python bench.py --gpus $N --placement tiers --tier-shapes tiny,tiny,tiny --dist-backend gloo \
    --steps 6 --warmup 2 > gpurun_out/r03_tiers_tiny_n${N}_gloo.json 2> gpurun_out/r03_tiers_tiny_n${N}_gloo.err
done
$T python bench.py --placement tiers --steps 8 --warmup 2 > gpurun_out/r03_tiers_real_n1.json
tail -c 3000 gpurun_out/r03_tiers_tiny_n1.json; echo; tail -c 1500 gpurun_out/r03_tiers_tiny_n2_gloo.json; echo; tail -c 1500 gpurun_out/r03_tiers_tiny_n4_gloo.json; echo
tail -c 4000 gpurun_out/r03_tiers_real_n1.json
