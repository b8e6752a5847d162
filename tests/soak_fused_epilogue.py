#!/usr/bin/env python3
"""Soak of the one-launch step (asd_verify_accept_fused_ex) against the two-launch step (asd_verify_accept_ex +
asd_predictor_stop) on random shapes -- every output bit for bit, the workspace handed back empty, the status word clean:

    python tests/soak_fused_epilogue.py [cases] [seed]

Draws B, K (1 ... 16 in-kernel; some longer drafts take the two-launch route inside the entry point), V, storage dtype,
temperature, the statistics column (incl. columns that straddle the two half-waves and -1 = no overlay), the hierarchy's depth,
stage, prefix rule, risk adjustment, theta, and the predictor shape (64 -> 32 -> 1 or 256 -> 128 -> 1).  Not collected by pytest
(a GPU soak, ~1 minute per 300 cases); the statistics of the last case are also compared with the oracle's numpy restatement."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from asd_amd import kernels as K  # noqa: E402
from oracle import oracle as O  # noqa: E402


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    dev = torch.device("cuda", 0)
    preds = {}
    for in_dim, hid in ((64, 32), (256, 128)):
        w1 = (rng.standard_normal((hid, in_dim)) / np.sqrt(in_dim)).astype(np.float32)
        b1 = (rng.standard_normal(hid) * 0.1).astype(np.float32)
        w2 = (rng.standard_normal((1, hid)) / np.sqrt(hid)).astype(np.float32)
        b2 = np.array([0.1], np.float32)
        preds[(in_dim, hid)] = (K.pack_mlp_weights(w1, b1, w2, b2), (w1, b1, w2, b2))
    in_kernel = 0
    cus = K.device_cu_count()
    for it in range(cases):
        Kd = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 8, 8, 9, 12, 15, 16, 16, 20, 33]))
        Bn = int(rng.choice([1, 2, 5, 8, 16, 31, 32, 33, 64, 100, 128, 160]))
        V = int(rng.choice([1000, 4000, 9999, 32000, 50257, 152064]))
        if Bn * Kd * V > 3.0e8:
            V = 9999
        dt = [torch.bfloat16, torch.float16, torch.float32][int(rng.integers(0, 3))]
        T = float(rng.choice([1.0, 0.7, 1.3]))
        in_dim, hid = [(64, 32), (256, 128)][int(rng.integers(0, 2))]
        packed, _ = preds[(in_dim, hid)]
        col = int(rng.choice([-1, 0, 5, 27, 28, 30, 31, 32, in_dim - 5]))
        L = int(rng.integers(1, 5))
        stage = int(rng.integers(0, L))
        prefix = bool(rng.integers(0, 2))
        risk = bool(rng.integers(0, 2))
        use_theta = bool(rng.integers(0, 2))
        want_stats = bool(rng.integers(0, 2)) or col >= 0
        g = torch.Generator(device=dev).manual_seed(int(rng.integers(0, 1 << 30)))
        lg = (torch.randn((Bn, Kd, V), generator=g, device=dev) * 4).to(dt)
        tok = torch.randint(0, V, (Bn, Kd), generator=g, device=dev, dtype=torch.int32)
        lp_d = -torch.rand((Bn, Kd), generator=g, device=dev) * 3
        u = torch.rand((Bn, Kd), generator=g, device=dev)
        feat = torch.randn((Bn, in_dim), generator=g, device=dev) * 0.3
        Cc = torch.tensor([1.0, 4.5, 10.0, 20.0][:L], dtype=torch.float64, device=dev)
        theta = torch.tensor([0.6, 0.4, 0.2, 0.0][:L], dtype=torch.float64, device=dev) if use_theta else None
        ws = K.VerifyWorkspace(Bn, Kd, V, dt)
        inv_t = float(np.float32(1.0 / T))
        ph0 = torch.rand((Bn, L), generator=g, device=dev, dtype=torch.float64) * 0.8 + 0.2
        kw = dict(stage_idx=stage, L=L, stats_col=col, risk_adjustment=risk, n_obs=120, alpha=1.0, beta=1.5, Cc=Cc, lam=0.8, theta=theta,
                  want_stats=want_stats, prefix_rule=prefix)
        for rep in range(2):
            ph1, ph2 = ph0.clone(), ph0.clone()
            v1 = K.verify_accept(lg, tok, lp_d, u, ws, inv_temperature=inv_t)
            s1 = K.predictor_stop(feat, packed, in_dim, hid, lp=v1.lp_target, p_hist=ph1, **kw)
            v2, s2 = K.verify_accept_fused(lg, tok, lp_d, u, ws, feat, packed, in_dim, hid, p_hist=ph2, inv_temperature=inv_t, **kw)
            torch.cuda.synchronize()
            ctx = (it, Bn, Kd, V, str(dt), T, in_dim, col, L, stage, prefix, risk, use_theta, want_stats)
            assert int(ws.buf.count_nonzero()) == 0, ("workspace not handed back empty", ctx)
            pairs = [("lp_t", v1.lp_target, v2.lp_target), ("accept", v1.accept, v2.accept), ("n_acc", v1.n_acc, v2.n_acc),
                     ("bits", v1.accept_bits, v2.accept_bits), ("score", s1.score, s2.score), ("k_star", s1.k_star, s2.k_star),
                     ("stop", s1.stop, s2.stop), ("thr_stop", s1.thr_stop, s2.thr_stop), ("p_hist", ph1, ph2), ("stats", s1.stats, s2.stats)]
            for name, a, b in pairs:
                if a is None and b is None:
                    continue
                assert torch.equal(a.view(torch.uint8) if a.is_floating_point() else a, b.view(torch.uint8) if b.is_floating_point() else b), (name, ctx)
        if want_stats:
            st = O.logprob_stats(v2.lp_target.cpu().numpy(), None, Kd)
            got = s2.stats.cpu().numpy()
            ok = np.isfinite(st).all(axis=1)
            assert got[ok].tobytes() == st[ok].tobytes(), ("statistics differ from numpy's", ctx)
        in_kernel += int(Kd <= 16 and L <= 4 and (in_dim == 64 or Bn * Kd >= cus))
        if (it + 1) % 50 == 0:
            print(f"{it + 1} cases ok ({in_kernel} of them through the in-kernel epilogue)", flush=True)
    print(f"soak ok: {cases} cases, {in_kernel} through the in-kernel epilogue")


if __name__ == "__main__":
    main()
