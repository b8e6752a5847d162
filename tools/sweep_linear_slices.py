"""X3: asd_linear with the reduction-slice count forced to 1 .. 32 on the decoder-layer matrices (graph-timed like
bench_linear.py): the measurements behind linear_plan's cost model.  `plan` marks the launcher's own choice.
    python tools/sweep_linear_slices.py --model 7b --rows 32 [--out gpurun_out/slices.json]"""
import argparse
import json
import sys
from importlib import import_module
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
K_ = import_module("adaptive-speculative-decoding_amd.kernels")
SL = import_module("adaptive-speculative-decoding_amd.serving.synthetic_lm")
BL = import_module("tools.bench_linear")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="7b")
    ap.add_argument("--rows", default="32")
    ap.add_argument("--slices", default="1,2,3,4,5,6,7,8,10,12,14,16,18,20,24,28,32")
    ap.add_argument("--out", default=None)
    ap.add_argument("--only", default=None, help="comma-separated matrix names (qkv, o, gate_up, down, lm_head)")
    a = ap.parse_args()
    lib = K_.test_hooks().__enter__()          # the TEST build of the library for the whole program (asd_debug_force_linear_slices)
    ws = K_.LinearWorkspace("cuda")
    s = SL.QWEN25_SHAPES[a.model]
    kv = s.kv_heads * s.head_dim
    mats = {"qkv": (s.hidden + 2 * kv, s.hidden), "o": (s.hidden, s.hidden), "gate_up": (2 * s.intermediate, s.hidden),
            "down": (s.hidden, s.intermediate), "lm_head": (s.vocab, s.hidden)}
    res = []
    for mname, (N, D) in mats.items():
        if a.only and mname not in a.only.split(","):
            continue
        wbytes = N * D * 2
        n_rot = max(2, min(8, (600 << 20) // wbytes + 1))
        W = [torch.randn(N, D, device="cuda", dtype=torch.bfloat16) * D ** -0.5 for _ in range(n_rot)]
        for M in [int(v) for v in a.rows.split(",")]:
            x = torch.randn(M, D, device="cuda", dtype=torch.bfloat16)
            out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
            plan = int(lib.asd_linear_slices(M, N, D))
            row = {"model": a.model, "matrix": mname, "M": M, "N": N, "D": D, "plan": plan, "us": {}}
            ws.buf = torch.empty(32 * M * N * 4 + 1024, dtype=torch.uint8, device="cuda")
            for k in [int(v) for v in a.slices.split(",")]:
                if k > D // 64:
                    continue
                lib.asd_debug_force_linear_slices(k)
                row["us"][k] = round(BL.time_us(lambda i: K_.linear(x, W[i], workspace=ws, out=out), n_rot), 2)
            lib.asd_debug_force_linear_slices(0)
            best = min(row["us"], key=row["us"].get)
            row["best"] = best
            res.append(row)
            print(mname, M, "plan", plan, "best", best, row["us"], flush=True)
        del W
        torch.cuda.empty_cache()
    if a.out:
        Path(a.out).write_text(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
