#!/usr/bin/env python3
"""N2 timing: asd_lm_head_verify (lm_head GEMM + LSE + accept, logits never in HBM) beside the
two-step path it replaces (torch / hipBLASLt bf16 GEMM that writes [B,K,V] logits, then
asd_verify_accept), on the Qwen2.5 lm_head shapes of configs/models.yaml.

    python tools/bench_lm_head.py [--out gpurun_out/lm_head.json] [--shapes 7b,32b,72b]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from asd_amd import kernels as K  # noqa: E402

SHAPES = {"7b": 3584, "14b": 5120, "32b": 5120, "72b": 8192}


def time_us(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = None
    for _ in range(3):
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        best = us if best is None else min(best, us)
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "lm_head.json"))
    ap.add_argument("--shapes", default="7b,32b,72b")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--draft-len", type=int, default=8)
    ap.add_argument("--vocab", type=int, default=152064)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--dtype", choices=["bf16", "f16"], default="bf16")
    a = ap.parse_args()
    B, Kk, V = a.batch, a.draft_len, a.vocab
    dt = torch.bfloat16 if a.dtype == "bf16" else torch.float16
    M = B * Kk
    rows = []
    for name in a.shapes.split(","):
        D = SHAPES[name]
        g = torch.Generator(device="cuda").manual_seed(1)
        h = torch.randn((M, D), device="cuda", generator=g).to(dt)
        w = (torch.randn((V, D), device="cuda", generator=g) * (3.0 / D ** 0.5)).to(dt)
        tok = torch.randint(0, V, (B, Kk), device="cuda", dtype=torch.int32)
        lp_d = -torch.rand((B, Kk), device="cuda")
        u = torch.rand((B, Kk), device="cuda")
        fused = K.LmHeadVerifier(w, B, Kk)
        out_f = fused(h, tok, lp_d, u)
        packed = K.LmHeadVerifier(w, B, Kk, packed=True)
        out_p = packed(h, tok, lp_d, u)
        ws = K.VerifyWorkspace(B, Kk, V, dt)
        logits = torch.empty((M, V), dtype=dt, device="cuda")
        out_t = K.verify_accept(logits.view(B, Kk, V), tok, lp_d, u, ws)

        def two_step():
            torch.matmul(h, w.t(), out=logits)
            K.verify_accept(logits.view(B, Kk, V), tok, lp_d, u, ws, out=out_t)

        t_fused = time_us(lambda: fused(h, tok, lp_d, u, out=out_f), a.reps)
        t_packed = time_us(lambda: packed(h, tok, lp_d, u, out=out_p), a.reps)
        t_gemm = time_us(lambda: torch.matmul(h, w.t(), out=logits), a.reps)
        t_two = time_us(two_step, a.reps)
        flops = 2.0 * M * D * V
        bytes_fused = V * D * 2 + M * D * 2
        bytes_two = bytes_fused + 2 * M * V * 2          # logits written once, read once
        row = dict(shape=name, D=D, V=V, B=B, K=Kk, fused_us=t_fused, fused_packed_us=t_packed, gemm_only_us=t_gemm,
                   two_step_us=t_two, fused_tflops=flops / t_fused / 1e6, fused_packed_tflops=flops / t_packed / 1e6,
                   fused_packed_hbm_gbs=bytes_fused / t_packed / 1e3, gemm_tflops=flops / t_gemm / 1e6,
                   fused_hbm_gbs=bytes_fused / t_fused / 1e3, two_step_hbm_gbs=bytes_two / t_two / 1e3,
                   speedup_vs_two_step=t_two / t_fused)
        rows.append(row)
        print(json.dumps(row), flush=True)
        del h, w, logits, fused, packed, ws
        torch.cuda.empty_cache()
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, "w") as f:
        json.dump(dict(device=torch.cuda.get_device_name(0), rows=rows), f, indent=1)


if __name__ == "__main__":
    main()
