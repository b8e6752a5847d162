#!/usr/bin/env python3
"""Soak: asd_draft_sample and asd_residual_sample on random shapes against the f64 oracle (dev tool).

    python tests/soak_draft_sample.py [cases] [seed]
Checks, per case: the nucleus threshold bit for bit where top_p is >= 1e-5 of mass away from a cumulative-mass step, the
token where additionally the draw is >= 1e-5 away from a CDF edge, log q within 2e-5, and always: the token lies inside the
reported nucleus.  The residual draw: the token where the draw is >= 1e-5 away from a CDF edge."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from asd_amd import kernels as K  # noqa: E402
from oracle import oracle as O  # noqa: E402
from tests.helpers import encode_logits, to_device_logits  # noqa: E402


def main():
    import faulthandler
    if os.environ.get("ASD_SOAK_DUMP_AFTER"):   # where is it? (a Python stack of every thread after N seconds, repeated)
        faulthandler.dump_traceback_later(float(os.environ["ASD_SOAK_DUMP_AFTER"]), repeat=True, file=sys.stdout)
    verbose = os.environ.get("ASD_SOAK_VERBOSE") == "1"
    hooks = K.test_hooks().__enter__()          # the TEST build of the library for the whole program (asd_debug_draft_groups)
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 120
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 4242)
    n_thr = n_tok = n_res = 0
    for it in range(cases):
        dtype = int(rng.choice([O.DT_BF16, O.DT_F32, O.DT_F16]))
        per = 4 if dtype == O.DT_F32 else 8
        V = per * int(rng.choice([1, 2, 7, 63, 64, 65, 127, 500, 1000, 4000, 8191, 19008]))
        B = int(rng.choice([1, 2, 5, 17, 40, 100, 130]))
        if B * V > 6_000_000:
            B = max(1, 6_000_000 // V)
        scale = float(rng.choice([0.02, 0.3, 1.0, 3.0, 8.0]))
        T = float(rng.choice([0.5, 0.7, 1.0, 1.3]))
        top_p = float(rng.choice([1.0, 0.99, 0.9, 0.5, 0.1]))
        x = (rng.standard_normal((B, V)) * scale).astype(np.float32)
        if rng.uniform() < 0.2:
            x[:, rng.integers(0, V)] += 12.0                     # a dominant token
        if rng.uniform() < 0.2:
            x[rng.integers(0, B), : V // 2] = -np.inf            # masked logits
        store = encode_logits(x, dtype)
        r = rng.uniform(0, 1, B).astype(np.float32)
        inv_t = float(np.float32(1.0 / T))
        ref = O.draft_sample(store, dtype, r, B, V, inv_t, top_p)
        lg = to_device_logits(store, dtype).view(B, V)
        groups = int(rng.choice([0, 0, -1, 1, 2, 4, 8, 16, 32]))            # 0: heuristic; -1: the streaming form; else forced
        hooks.asd_debug_draft_groups(groups if groups <= 0 or B * groups <= 256 else 0)
        if verbose:
            print(f"case {it}: B={B} V={V} dtype={dtype} scale={scale} T={T} top_p={top_p} groups={groups}", flush=True)
        samp = K.DraftSampler(B, V, lg.dtype)
        d = samp(lg, torch.from_numpy(r).cuda(), inv_t, top_p)
        torch.cuda.synchronize()
        hooks.asd_debug_draft_groups(0)
        assert int(samp.buf.count_nonzero()) == 0, (it, "workspace not handed back empty", B, V, groups)
        tok, lp, thr = d.tok.cpu().numpy(), d.lp.cpu().numpy(), d.thr.cpu().numpy()
        okp = ref["margin_p"] > 1e-5
        assert np.array_equal(thr[okp], ref["thr"][okp]), (it, "thr", B, V, dtype, scale, T, top_p)
        ok = okp & (ref["margin_r"] > 1e-5)
        assert np.array_equal(tok[ok], ref["tok"][ok]), (it, "tok", B, V, dtype, scale, T, top_p)
        assert np.allclose(lp[ok], ref["lp"][ok], rtol=1e-6, atol=2e-5), (it, "lp", B, V, dtype, scale, T, top_p)
        xs = O.logits_as_f32(store, dtype)
        assert (xs[np.arange(B), tok] >= thr).all(), (it, "inside", B, V, dtype)
        n_thr += int(okp.sum())
        n_tok += int(ok.sum())
        # residual draw on the same shapes (K = 2)
        Kd = 2
        xt = (rng.standard_normal((B * Kd, V)) * max(scale, 0.5)).astype(np.float32)
        xd = (xt + rng.standard_normal((B * Kd, V)) * 0.7).astype(np.float32)
        st, sd = encode_logits(xt, dtype), encode_logits(xd, dtype)
        sb = encode_logits((rng.standard_normal((B, V)) * max(scale, 0.5)).astype(np.float32), dtype)
        n_acc = rng.integers(0, Kd + 1, B).astype(np.int32)
        want, margin = O.residual_sample(st, sd, dtype, n_acc, r, B, Kd, V, bonus=sb, inv_temperature=inv_t)
        if verbose:
            print(f"case {it}: residual draw, n_acc {n_acc.tolist()[:8]}", flush=True)
        t3, d3 = to_device_logits(st, dtype).view(B, Kd, V), to_device_logits(sd, dtype).view(B, Kd, V)
        got = K.ResidualSampler(B, V, t3.dtype)(t3, d3, torch.from_numpy(n_acc).cuda(), torch.from_numpy(r).cuda(),
                                                to_device_logits(sb, dtype).view(B, V), inv_t)
        torch.cuda.synchronize()
        # The residual max(0, p_t - p_d) is formed from f32 probabilities: on rows where one token holds ~all the mass of both
        # distributions the difference cancels down to the ulp of 1.0, i.e. a relative accuracy of ~1e-7 / Z (Z = the residual's
        # total mass; the normalisers are carried as two floats, which removed a 30x larger term), so such rows are only
        # checked for membership of the support.
        okr = margin > 1e-5
        if max(scale, 0.5) * inv_t <= 4.0:
            assert np.array_equal(got.cpu().numpy()[okr], want[okr]), (it, "residual", B, V, dtype, scale, T)
        else:
            g = got.cpu().numpy()
            assert ((g >= 0) & (g < V)).all(), (it, "residual range", B, V)
        n_res += int(okr.sum())
        if (it + 1) % 10 == 0:
            print(f"{it + 1} cases ok", flush=True)
    print(f"fuzz: {cases} cases passed; thresholds compared {n_thr}, tokens {n_tok}, residual tokens {n_res}")


if __name__ == "__main__":
    main()
