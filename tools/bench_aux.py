#!/usr/bin/env python3
"""Timing of the decision / predictor kernels (SURVEY §8a rows A1, A2, A7, A8, A11 and the fused
epilogue) at a latency batch (B = 32, what one decode step sees) and a throughput batch, with the
CPU numbers beside them:  `cpu_oracle` = oracle/asd_oracle.c (one thread),  `cpu_python` = the
reference's own idiom restated in pure Python / torch-CPU (oracle.py: py_*), on a bounded sample.

    python tools/bench_aux.py [--out gpurun_out/aux_kernels.json]

GPU time = HIP events around `reps` back-to-back launches on the current stream (includes launch
gaps; for the B = 32 rows that IS the quantity of interest).  GB/s = algorithmic bytes / time.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from asd_amd import kernels as K  # noqa: E402
from oracle import oracle as O  # noqa: E402


def gpu_us(fn, reps):
    for _ in range(5):
        fn()
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


def cpu_us(fn, budget_s=1.0):
    fn()
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < budget_s:
        fn()
        n += 1
    return (time.perf_counter() - t0) * 1e6 / n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "aux_kernels.json"))
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(0)
    rows = []

    def record(name, B, gpu, nbytes, cpu_oracle_us, cpu_python_us_per_item, note=""):
        rows.append(dict(kernel=name, batch=B, gpu_us=gpu, gpu_gbs=nbytes / gpu / 1e3, algorithmic_bytes=nbytes,
                         cpu_oracle_us=cpu_oracle_us, cpu_python_us_per_item=cpu_python_us_per_item,
                         gpu_items_per_s=B / gpu * 1e6, note=note))
        print(f"{name:28s} B={B:8d}  gpu {gpu:9.2f} us  {nbytes / gpu / 1e3:8.1f} GB/s   cpu-oracle {cpu_oracle_us:10.1f} us"
              f"   python/item {cpu_python_us_per_item:7.2f} us  {note}", flush=True)

    L = 4
    Cn = np.array([1.0, 1.6, 4.2, 8.8])
    Cc = torch.from_numpy(Cn).to(dev)
    # ---------------- A1 optimal_stopping_rule
    py_dp = cpu_us(lambda: O.py_optimal_stopping_rule([0.3, 0.5, 0.8, 1.0], [1.0, 1.6, 4.2, 8.8], 1.0), 0.5)
    for B in (32, 1 << 20):
        pn = rng.uniform(0, 1, (B, L))
        p = torch.from_numpy(pn).to(dev)
        g = gpu_us(lambda: K.optimal_stopping(p, Cc, 1.0), 200 if B == 32 else 50)
        c = cpu_us(lambda: O.optimal_stopping(pn, Cn, 1.0), 0.5)
        record("A1 asd_optimal_stopping", B, g, B * (L * 8 + 4 + (L + 1) * 8), c, py_dp, "L=4, J written")
    # ---------------- A2 bayesian_adjustment
    py_b = cpu_us(lambda: O.py_bayesian_adjustment(0.4, 100, 1.0, 1.0), 0.3)
    for B in (32, 1 << 22):
        pn = rng.uniform(0, 1, B)
        p = torch.from_numpy(pn).to(dev)
        g = gpu_us(lambda: K.bayes_adjust(p, 100), 200 if B == 32 else 50)
        c = cpu_us(lambda: O.bayes_adjust(pn, 100), 0.5)
        record("A2 asd_bayes_adjust", B, g, B * 16, c, py_b)
    # ---------------- A8 predictor MLP
    w1 = (rng.standard_normal((32, 64)) / 8).astype(np.float32)
    b1 = np.zeros(32, np.float32)
    w2 = (rng.standard_normal((1, 32)) / 6).astype(np.float32)
    b2 = np.zeros(1, np.float32)
    packed = K.pack_mlp_weights(w1, b1, w2, b2)
    mod = torch.nn.Sequential(torch.nn.Linear(64, 32), torch.nn.ReLU(), torch.nn.Dropout(0.1), torch.nn.Linear(32, 1),
                              torch.nn.Sigmoid()).eval()
    x1 = torch.zeros(1, 64)
    with torch.no_grad():
        py_mlp = cpu_us(lambda: mod(x1).item(), 0.5)        # minimal_adaptive_decoder.py:156 idiom
    for B in (32, 1 << 20):
        xn = rng.standard_normal((B, 64)).astype(np.float32)
        x = torch.from_numpy(xn).to(dev)
        g = gpu_us(lambda: K.mlp_predict(x, packed, 64, 32), 200 if B == 32 else 20)
        c = cpu_us(lambda: O.mlp_predict(xn, w1, b1, w2[0], b2), 0.5)
        record("A8 asd_mlp_predict 64x32x1", B, g, B * (64 * 4 + 4), c, py_mlp, "torch-CPU .item() per row as python idiom")
    # ---------------- A11 threshold stop
    theta = torch.tensor([0.636, 0.48, 0.2258, 0.0], dtype=torch.float64, device=dev)
    for B in (32, 1 << 22):
        sn = rng.uniform(0, 1, B).astype(np.float32)
        sc = torch.from_numpy(sn).to(dev)
        g = gpu_us(lambda: K.threshold_stop(sc, theta), 200 if B == 32 else 50)
        c = cpu_us(lambda: O.threshold_stop(sn, theta.cpu().numpy()), 0.5)
        record("A11 asd_threshold_stop", B, g, B * 8, c, 0.25, "python idiom ~0.25 us (4 compares)")
    # ---------------- A7 log-prob statistics
    lp8 = [float(v) for v in -np.abs(rng.standard_normal(8))]
    lp128 = [float(v) for v in -np.abs(rng.standard_normal(128))]
    py_s8 = cpu_us(lambda: O.py_logprob_stats(lp8), 0.5)
    py_s128 = cpu_us(lambda: O.py_logprob_stats(lp128), 0.5)
    for B, Kk, py in ((32, 8, py_s8), (32, 128, py_s128), (1 << 16, 8, py_s8), (1 << 14, 128, py_s128)):
        ln = (-np.abs(rng.standard_normal((B, Kk)))).astype(np.float32)
        lpt = torch.from_numpy(ln).to(dev)
        g = gpu_us(lambda: K.logprob_stats(lpt), 200 if B == 32 else 20)
        c = cpu_us(lambda: O.logprob_stats(ln, None, Kk), 0.5)
        record(f"A7 asd_logprob_stats K={Kk}", B, g, B * (Kk * 4 + 40), c, py, "python idiom = 5 numpy calls per sequence")
    # ---------------- fused epilogue
    for B in (32, 128, 8192):
        feat = torch.from_numpy(rng.standard_normal((B, 64)).astype(np.float32)).to(dev)
        lpt = torch.from_numpy((-np.abs(rng.standard_normal((B, 8)))).astype(np.float32)).to(dev)
        ph = torch.ones((B, 3), dtype=torch.float64, device=dev)
        C3 = torch.tensor([1.0, 4.5, 10.0], dtype=torch.float64, device=dev)
        g = gpu_us(lambda: K.predictor_stop(feat, packed, 64, 32, stage_idx=0, L=3, lp=lpt, stats_col=5, p_hist=ph, Cc=C3,
                                            lam=1.0), 200 if B <= 128 else 50)
        record("N1 asd_predictor_stop", B, g, B * (64 * 4 + 32 + 24 + 4 + 4 + 1), float("nan"),
               py_s8 + py_mlp + py_b + py_dp, "python idiom = stats + MLP + Bayes + DP per request")
    # ---------------- commit step: residual / bonus draw
    for B, Kk, V in ((32, 8, 152064), (128, 8, 152064)):
        t = (torch.randn((B, Kk, V), device=dev) * 3).to(torch.bfloat16)
        d = (t.float() + torch.randn((B, Kk, V), device=dev)).to(torch.bfloat16)
        bonus = (torch.randn((B, V), device=dev) * 3).to(torch.bfloat16)
        n_acc = torch.randint(0, Kk + 1, (B,), device=dev, dtype=torch.int32)
        r = torch.rand((B,), device=dev)
        samp = K.ResidualSampler(B, V)
        out = torch.empty((B,), dtype=torch.int32, device=dev)
        g = gpu_us(lambda: samp(t, d, n_acc, r, bonus, 1.0, out), 100)
        record("asd_residual_sample", B, g, 2 * B * V * 2, float("nan"), float("nan"),
               "3 launches; bytes = one pass over the B target + B draft rows (second pass is L2 / MALL traffic)")
    # ---------------- N3 commit / KV-rollback bookkeeping
    for B in (32, 4096):
        Kk, T = 8, 512
        tok = torch.randint(0, 1000, (B, Kk), device=dev, dtype=torch.int32)
        n_acc = torch.randint(0, Kk + 1, (B,), device=dev, dtype=torch.int32)
        drawn = torch.randint(0, 1000, (B,), device=dev, dtype=torch.int32)
        seq_len = torch.full((B,), 16, dtype=torch.int32, device=dev)
        out_tok = torch.zeros((B, T), dtype=torch.int32, device=dev)
        nc = torch.empty((B,), dtype=torch.int32, device=dev)

        def commit():
            seq_len.fill_(16)                              # keep the rows from filling up across repetitions
            K.commit_step(tok, n_acc, drawn, seq_len, out_tok, nc)
        g = gpu_us(commit, 200)
        tn, nn_, dn = tok.cpu().numpy(), n_acc.cpu().numpy(), drawn.cpu().numpy()
        ln, on = np.full(B, 16, np.int32), np.zeros((B, T), np.int32)
        c = cpu_us(lambda: O.commit_step(tn, nn_, dn, ln, on), 0.5)
        record("N3 asd_commit_step", B, g, B * (Kk * 4 + 4 + 4 + 4 + (Kk + 1) * 4 + 4), c, c / B,
               "gpu time includes the seq_len reset fill; cpu = numpy/python oracle loop")
    # ---------------- N4 lambda sweep
    for B, G in ((96, 20), (65536, 64)):
        pn = np.sort(rng.uniform(0.2, 1.0, (B, 4)), axis=1)
        lam_n = np.logspace(-2, 2, G)
        p_, l_ = torch.from_numpy(pn).to(dev), torch.from_numpy(lam_n).to(dev)
        C4 = torch.tensor([1.0, 1.6, 4.2, 8.8], dtype=torch.float64, device=dev)
        g = gpu_us(lambda: K.lambda_sweep(p_, C4, l_), 200 if B == 96 else 20)
        nb = min(B, 96)
        c = cpu_us(lambda: O.lambda_sweep(pn[:nb], Cn, lam_n[:4]), 0.5) * (B / nb) * (G / 4)
        record("N4 asd_lambda_sweep", B * G, g, B * 32 + G * 8 + B * G * 20, c, py_dp,
               f"{G} lambdas x {B} requests in one launch; cpu = oracle scaled from a {nb} x 4 sample; python idiom = one DP call")
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, "w") as f:
        json.dump(dict(device=torch.cuda.get_device_name(0), rows=rows), f, indent=1)


if __name__ == "__main__":
    main()
