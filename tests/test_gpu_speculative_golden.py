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
from tests.helpers import run_gpu_verify, spec_cases, spec_full_cases, to_device_logits

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


def test_accept_residual_and_proposal_at_the_full_vocabulary_against_hf(golden):
    """the kernels against HF at V = 152064 on bf16- and f16-stored rows with the reference's sampling
    settings (T = 0.7 folded into the kernels' FMA constant, top-p 0.9 on the draft side).  16 cases
    (tests/golden/speculative_sampling_full.npz): asd_verify_accept_ex gives HF's n_matches exactly; asd_residual_sample_ex
    (x* thresholds: the residual against the nucleus-truncated draft row) gives the token the inverse CDF of HF's p' gives
    wherever the draw is >= 1e-5 of the mass from a CDF edge; asd_draft_sample reproduces HF's nucleus threshold bit for
    bit, the drafted token and log q(token)."""
    import torch

    from asd_amd import kernels as K_
    g = golden.npz("speculative_sampling_full.npz")
    n_cases = n_draws = n_prop = 0
    for c in spec_full_cases(g):
        K, V, dt = c["K"], c["V"], c["dtype"]
        t = to_device_logits(c["new"][:K], dt).view(1, K, V)
        d = to_device_logits(c["cand"], dt).view(1, K, V)
        bo = to_device_logits(c["new"][K:K + 1], dt).view(1, V)
        ws = K_.VerifyWorkspace(1, K, V, t.dtype)
        v = K_.verify_accept(t, torch.from_numpy(c["tok"]).cuda().view(1, K), torch.from_numpy(c["lq"].astype(np.float32)).cuda().view(1, K),
                             torch.from_numpy(c["u"]).cuda().view(1, K), ws, inv_temperature=c["inv_t"])
        torch.cuda.synchronize()
        assert int(v.n_acc.cpu()[0]) == c["n_matches"], c["case"]
        # (HF's warper drops some of the scores EQUAL to the nucleus threshold, the kernels keep every tie: see the CPU twin)
        same_residual = c["n_matches"] == K or c["ties_removed"][c["n_matches"]] == 0
        samp = K_.ResidualSampler(1, V, t.dtype)
        thr = torch.from_numpy(c["thr"]).cuda().view(1, K)
        for r, want, margin in zip(c["r"], c["want_tok"], c["margin"]):
            tok = samp(t, d, torch.tensor([c["n_matches"]], dtype=torch.int32, device="cuda"),
                       torch.tensor([float(r)], dtype=torch.float32, device="cuda"), bo, c["inv_t"], d_threshold=thr)
            torch.cuda.synchronize()
            if margin > 1e-5 and same_residual:
                assert int(tok.cpu()[0]) == int(want), c["case"]
                n_draws += 1
        # the proposal step on the K draft rows as one batch
        dd = K_.DraftSampler(K, V, d.dtype)(d.view(K, V), torch.from_numpy(c["pick"]).cuda(), c["inv_t"], c["top_p"])
        torch.cuda.synchronize()
        ref = O.draft_sample(c["cand"], dt, c["pick"], K, V, c["inv_t"], c["top_p"])
        ok = ref["margin_p"] > 1e-5
        assert np.array_equal(dd.thr.cpu().numpy()[ok], c["thr"][ok]), c["case"]
        okt = ok & (c["pick_margin"] > 1e-5) & (ref["margin_r"] > 1e-5) & (c["ties_removed"] == 0)   # same nucleus as HF's
        assert np.array_equal(dd.tok.cpu().numpy()[okt], c["tok"][okt]), c["case"]
        np.testing.assert_allclose(dd.lp.cpu().numpy()[okt], c["lq"][okt], rtol=0, atol=2e-5)
        n_prop += int(okt.sum())
        n_cases += 1
    assert n_cases == 16 and n_draws >= 24 and n_prop >= 24
