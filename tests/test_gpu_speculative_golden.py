"""asd_verify_accept + asd_residual_sample against transformers' `_speculative_sampling` (HF assisted generation: the
accept test r_i <= p_i / q_i, n_matches = the leading accepted run, the residual distribution norm(max(0, p - q)) and the
bonus distribution p_{n+1}) on tests/golden/speculative_sampling.npz -- the function was called unmodified in the dev
container with its two random draws supplied / recorded from outside (oracle/gen_golden.py::gen_speculative_sampling).
The reference has no token-level accept test; this pins the build's A5 to the implementation its ecosystem
uses.  Bars: n_acc equal on every case (the fixture's uniforms keep 1e-3 away from the decision edge); the drawn token equal
wherever the draw is >= 1e-5 of the mass away from a CDF edge."""
import numpy as np
import pytest

from oracle import oracle as O
from tests.helpers import run_gpu_verify, spec_cases, to_device_logits

pytestmark = pytest.mark.gpu


def test_accept_and_residual_against_hf_speculative_sampling(golden):
    import torch

    from asd_amd import kernels as K_
    g = golden.npz("speculative_sampling.npz")
    n_cases = n_draws = 0
    for c in spec_cases(g):
        K, V = c["K"], c["V"]
        case = dict(logits=c["new"][:K], dtype=O.DT_F32, B=1, K=K, V=V, ld=V, tok=c["tok"].reshape(1, K),
                    lp_d=c["lp_d"].reshape(1, K), u=c["u"].reshape(1, K))
        got = run_gpu_verify(case)
        assert int(got["n_acc"][0]) == c["n_matches"]
        t = to_device_logits(c["new"][:K], O.DT_F32).view(1, K, V)
        d = to_device_logits(c["cand"], O.DT_F32).view(1, K, V)
        bo = to_device_logits(c["new"][K:K + 1], O.DT_F32).view(1, V)
        samp = K_.ResidualSampler(1, V, t.dtype)
        for r, want, margin in zip(c["r"], c["want_tok"], c["margin"]):
            tok = samp(t, d, torch.tensor([c["n_matches"]], dtype=torch.int32, device="cuda"),
                       torch.tensor([float(r)], dtype=torch.float32, device="cuda"), bo, 1.0)
            torch.cuda.synchronize()
            if margin > 1e-5:
                assert int(tok.cpu()[0]) == int(want)
                n_draws += 1
        n_cases += 1
    assert n_cases == 24 and n_draws >= 60
