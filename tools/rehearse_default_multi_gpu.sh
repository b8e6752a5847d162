#!/bin/bash
# Rehearsal of the DEFAULT multi-GPU bench line (the bare `python bench.py --gpus N`, what the driver runs) on ONE GPU: the ranks
# share cuda:0, control and data path over gloo (device tensors staged through the host).  N = 2 runs real Qwen2.5 shapes one size
# down (7B / 14B / 32B as {7B + 14B | 32B}: two processes with the 7B / 32B / 72B set do not fit the ONE GPU's 288 GB next to their
# two HIP contexts and allocator reserves -- on two GPUs each rank has its own 288 GB); N = 4 (7B | 32B | 72B vocab-sharded over two
# ranks) would need two 72B bodies and runs tiny shapes.  The numbers are not scaling measurements: they show that the N > 1 branch -- headline, sharded_verify, the
# bounded `loop` sub-record of BASELINE configs[3] with its roofline / bytes_sent / rccl_ranks, the watchdog -- executes.
# Then the configs[4] loop (replicated drafts + sharded target) at 1 rank with the real 7B + 72B pair and at 2 / 4 ranks, tiny.
set -e -o pipefail
mkdir -p gpurun_out
R=${ROUND:-r04}
T="timeout -k 10 560"
ASD_BENCH_ONE_DEVICE=1 $T python bench.py --gpus 2 --dist-backend gloo --steps 20 --warmup 5 --loop-steps 6 --tier-shapes 7b,14b,32b \
  > gpurun_out/${R}_default_n2_gloo.json 2> gpurun_out/${R}_default_n2_gloo.err
ASD_BENCH_ONE_DEVICE=1 $T python bench.py --gpus 4 --dist-backend gloo --steps 20 --warmup 5 --loop-steps 6 --tier-shapes tiny,tiny,tiny \
  > gpurun_out/${R}_default_n4_gloo.json 2> gpurun_out/${R}_default_n4_gloo.err
$T python bench.py --placement sharded-target --tier-shapes 7b,72b --steps 6 --warmup 2 > gpurun_out/${R}_sharded_real_n1.json 2> gpurun_out/${R}_sharded_real_n1.err
for N in 2 4; do
  ASD_BENCH_ONE_DEVICE=1 $T python bench.py --gpus $N --placement sharded-target --tier-shapes tiny,tiny --dist-backend gloo \
    --steps 6 --warmup 2 > gpurun_out/${R}_sharded_tiny_n${N}_gloo.json 2> gpurun_out/${R}_sharded_tiny_n${N}_gloo.err
done
python - <<PY
import json
for f in ("${R}_default_n2_gloo", "${R}_default_n4_gloo"):
    d = json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1])
    l = d.get("loop") or {}
    print(f, d["n_gpus"], round(d["value"]), d["unit"], "| loop:", l.get("kind"), l.get("placement"), "rccl_ranks", l.get("rccl_ranks"), l.get("backend"),
          round(l.get("verified_tokens_per_s") or 0, 1), "tok/s", round(l.get("ms_per_step") or 0, 2), "ms/step", "roofline.frac", (l.get("roofline") or {}).get("frac"),
          "messages", l.get("messages_sent"), "error", l.get("error"))
for f in ("${R}_sharded_real_n1", "${R}_sharded_tiny_n2_gloo", "${R}_sharded_tiny_n4_gloo"):
    d = json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1])
    l = d["loop"]
    print(f, d["n_gpus"], round(d["value"], 1), "tok/s", round(l["ms_per_step"], 2), "ms/step", "batch", l["batch_total"], "fed/rank/step", l["fed_tokens_per_rank_per_step"],
          int(l["bytes_exchanged_per_step_rank0"]), "B/step", "roofline.frac", l["roofline"]["frac"])
PY
