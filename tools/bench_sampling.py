#!/usr/bin/env python3
"""Timing of the proposal / commit draws: asd_draft_sample (X1) and asd_residual_sample_ex, beside the torch idiom
they replace (log_softmax + multinomial + gather per drafted token, serving/speculative.py of round 1).

    python tools/bench_sampling.py [--out gpurun_out/sampling.json]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from asd_amd import kernels as K  # noqa: E402


def timed(fn, reps=200, settle=50):
    for _ in range(settle):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps      # us per call


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "sampling.json"))
    ap.add_argument("--reps", type=int, default=200)
    ap.add_argument("--batches", default="8,32,128")
    ap.add_argument("--scale", type=float, default=3.0, help="logits = scale * N(0,1): 3 = a wide nucleus (hundreds of tokens at "
                    "T = 0.7, top-p 0.9), 8 = a peaked row (a handful of tokens), closer to a confident LLM step")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    V = 152064
    res = {}
    for B in [int(x) for x in a.batches.split(",")]:
        g = torch.Generator(device=dev).manual_seed(B)
        nb = 8
        rows = [(torch.randn((B, V), generator=g, device=dev) * a.scale).to(torch.bfloat16) for _ in range(nb)]
        r = torch.rand((B,), generator=g, device=dev)
        ds = K.DraftSampler(B, V, torch.bfloat16, dev)
        out = None
        i = [0]

        def draft(top_p):
            nonlocal out
            i[0] += 1
            out = ds(rows[i[0] % nb], r, 1 / 0.7, top_p, out)

        def torch_idiom():
            i[0] += 1
            lp = torch.log_softmax(rows[i[0] % nb].float() / 0.7, dim=-1)
            tok = torch.multinomial(lp.exp(), 1)[:, 0]
            return lp.gather(1, tok[:, None])

        Kd = 8
        t3 = torch.stack([rows[j % nb] for j in range(Kd)], 1).contiguous()
        d3 = torch.stack([rows[(j + 3) % nb] for j in range(Kd)], 1).contiguous()
        n_acc = torch.randint(0, Kd + 1, (B,), generator=g, device=dev, dtype=torch.int32)
        rs = K.ResidualSampler(B, V, torch.bfloat16, dev)
        thr = torch.full((B, Kd), 2.0, device=dev)
        tok_out = torch.empty((B,), dtype=torch.int32, device=dev)
        res[f"B{B}"] = {
            "asd_draft_sample_top_p_0.9_us": timed(lambda: draft(0.9), a.reps),
            "asd_draft_sample_no_top_p_us": timed(lambda: draft(1.0), a.reps),
            "torch_log_softmax_multinomial_gather_us": timed(torch_idiom, a.reps),
            "asd_residual_sample_us": timed(lambda: rs(t3, d3, n_acc, r, rows[0], 1 / 0.7, out=tok_out), a.reps),
            "asd_residual_sample_ex_truncated_draft_us": timed(lambda: rs(t3, d3, n_acc, r, rows[0], 1 / 0.7, out=tok_out, d_threshold=thr), a.reps),
            "row_bytes": V * 2, "rows": B,
        }
        print(B, res[f"B{B}"], flush=True)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
