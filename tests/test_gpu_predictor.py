"""Predictor-side kernels (A7 stats, A8 MLP, fused epilogue) against goldens and the oracle."""
import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu

SCORE_ATOL = 1e-5      # BASELINE.json: stopping scores within 1e-5 fp32 (observed ~1e-7)


@pytest.fixture(scope="module")
def K_():
    from asd_amd import kernels
    return kernels


def _cuda(a, dtype):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a, dtype=dtype)).cuda()


def test_mlp_golden_scores(golden, K_):
    g = golden.npz("predictor.npz")
    packed = K_.pack_mlp_weights(g["w1"], g["b1"], g["w2"], g["b2"])
    got = K_.mlp_predict(_cuda(g["X"], np.float32), packed, 64, 32).cpu().numpy()
    np.testing.assert_allclose(got, g["scores"], rtol=0, atol=SCORE_ATOL)
    np.testing.assert_allclose(got, g["scores_one_by_one"], rtol=0, atol=SCORE_ATOL)
    print("max |score_gpu - score_torch| =", np.abs(got - g["scores"]).max())


@pytest.mark.parametrize("D,H,B", [(64, 32, 1), (64, 32, 100003), (256, 128, 777), (5, 3, 9), (1024, 1024, 5),
                                   (100, 65, 130)])
def test_mlp_shapes_vs_oracle(K_, D, H, B):
    rng = np.random.default_rng(D * 1000 + H)
    w1 = (rng.standard_normal((H, D)) / np.sqrt(D)).astype(np.float32)
    b1 = rng.standard_normal(H).astype(np.float32) * 0.1
    w2 = (rng.standard_normal(H) / np.sqrt(H)).astype(np.float32)
    b2 = rng.standard_normal(1).astype(np.float32)
    x = rng.standard_normal((B, D)).astype(np.float32)
    packed = K_.pack_mlp_weights(w1, b1, w2[None, :], b2)
    got = K_.mlp_predict(_cuda(x, np.float32), packed, D, H).cpu().numpy()
    want = O.mlp_predict(x, w1, b1, w2, b2)
    np.testing.assert_allclose(got, want, rtol=0, atol=SCORE_ATOL)


def test_mlp_throughput_form_and_strided_rows(golden, K_):
    """B >= 4096 takes the lane-per-row kernel (weights on the scalar path); rows may be strided and
    need not be 16-byte aligned."""
    import torch
    g = golden.npz("predictor.npz")
    packed = K_.pack_mlp_weights(g["w1"], g["b1"], g["w2"], g["b2"])
    rng = np.random.default_rng(4)
    for B in (4096, 4097, 50000):
        big = rng.standard_normal((B, 67)).astype(np.float32)
        xb = torch.from_numpy(big).cuda()
        for view, ref in ((xb[:, :64], big[:, :64]), (xb[:, 1:65], big[:, 1:65]), (xb[:, 3:67].contiguous(), big[:, 3:67])):
            got = K_.mlp_predict(view, packed, 64, 32).cpu().numpy()
            want = O.mlp_predict(np.ascontiguousarray(ref), g["w1"], g["b1"], g["w2"][0], g["b2"])
            np.testing.assert_allclose(got, want, rtol=0, atol=SCORE_ATOL)
    small = K_.mlp_predict(torch.from_numpy(g["X"]).cuda(), packed, 64, 32).cpu().numpy()       # latency form
    rep = K_.mlp_predict(torch.from_numpy(np.tile(g["X"], (20, 1))).cuda(), packed, 64, 32).cpu().numpy()  # 5120 rows
    np.testing.assert_allclose(rep[:256], small, rtol=0, atol=1e-6)
    np.testing.assert_allclose(rep[:256], g["scores"], rtol=0, atol=SCORE_ATOL)


def test_logprob_stats_goldens_bit_exact(golden, K_):
    """A7 columns [5:10] of the reference's extract_features, float64, bit for bit."""
    g = golden.npz("features_a7.npz")
    lp = g["logprobs"].astype(np.float32)
    got = K_.logprob_stats(_cuda(lp, np.float32), _cuda(g["n_valid"], np.int32)).cpu().numpy()
    want = np.ascontiguousarray(g["features"][:, 5:10])
    assert got.tobytes() == want.tobytes()


@pytest.mark.parametrize("K", [1, 2, 7, 8, 9, 64, 129, 300, 1024])
def test_logprob_stats_vs_numpy(K_, K):
    rng = np.random.default_rng(K)
    B = 37
    lp = (-np.abs(rng.standard_normal((B, K))) * 3).astype(np.float32)
    lp[0, : min(K, 3)] = lp[0, 0]                      # ties
    nv = rng.integers(0, K + 1, B).astype(np.int32)
    nv[1] = K
    nv[2] = 0
    got = K_.logprob_stats(_cuda(lp, np.float32), _cuda(nv, np.int32)).cpu().numpy()
    for b in range(B):
        want = O.py_logprob_stats([float(v) for v in lp[b, :nv[b]]])
        assert got[b].tolist() == want, (b, nv[b])
    full = K_.logprob_stats(_cuda(lp, np.float32)).cpu().numpy()
    assert full[5].tolist() == O.py_logprob_stats([float(v) for v in lp[5]])


def test_fused_epilogue_equals_composition(golden, K_):
    """asd_predictor_stop == stats -> feature overlay -> MLP -> Bayes -> DP rule / theta test."""
    import torch
    g = golden.npz("predictor.npz")
    rng = np.random.default_rng(8)
    B, K, L = 200, 8, 3
    packed = K_.pack_mlp_weights(g["w1"], g["b1"], g["w2"], g["b2"])
    feat = rng.standard_normal((B, 64)).astype(np.float32) * 0.3
    lp = (-np.abs(rng.standard_normal((B, K))) * 2).astype(np.float32)
    nv = rng.integers(0, K + 1, B).astype(np.int32)
    Cc = np.array([1.0, 4.5, 10.0])
    theta, _ = O.derive_thresholds([0.7, 0.85, 0.9], Cc, 0.1)
    for stage_idx in (0, 1, 2):
        for prefix in (False, True):
            for risk in (False, True):
                p_hist = rng.uniform(0.2, 1.0, (B, L))
                p_hist[:, L - 1] = 1.0
                ph = _cuda(p_hist, np.float64)
                r = K_.predictor_stop(_cuda(feat, np.float32), packed, 64, 32, stage_idx=stage_idx, L=L,
                                      lp=_cuda(lp, np.float32), n_valid=_cuda(nv, np.int32), stats_col=5,
                                      risk_adjustment=risk, n_obs=150, alpha=1.0, beta=2.0, p_hist=ph,
                                      Cc=_cuda(Cc, np.float64), lam=0.7, prefix_rule=prefix,
                                      theta=_cuda(theta, np.float64), want_stats=True)
                torch.cuda.synchronize()
                stats = O.logprob_stats(lp, nv, K)
                assert r.stats.cpu().numpy().tobytes() == stats.tobytes()
                x = feat.copy()
                x[:, 5:10] = stats.astype(np.float32)
                score = O.mlp_predict(x, g["w1"], g["b1"], g["w2"][0], g["b2"])
                got_score = r.score.cpu().numpy()
                np.testing.assert_allclose(got_score, score, rtol=0, atol=SCORE_ATOL)
                # decisions are checked on the GPU's own score (bit-exact chain from there on)
                prob = got_score.astype(np.float64)
                if risk:
                    prob = O.bayes_adjust(prob, 150, 1.0, 2.0)
                want_hist = p_hist.copy()
                want_hist[:, stage_idx] = prob
                assert ph.cpu().numpy().tobytes() == want_hist.tobytes()
                n_dp = stage_idx + 1 if prefix else L
                ks, _ = O.optimal_stopping(want_hist[:, :n_dp], Cc[:n_dp], 0.7)
                assert np.array_equal(r.k_star.cpu().numpy(), ks)
                assert np.array_equal(r.stop.cpu().numpy(), (ks == stage_idx).astype(np.uint8))
                thr = ((got_score.astype(np.float64) >= theta[stage_idx]) | (stage_idx == L - 1)).astype(np.uint8)
                assert np.array_equal(r.thr_stop.cpu().numpy(), thr)
                if prefix and stage_idx == 0:
                    assert bool((r.stop == 1).all())       # the prefix rule always stops at stage 0


def test_fused_epilogue_without_stats(golden, K_):
    g = golden.npz("predictor.npz")
    packed = K_.pack_mlp_weights(g["w1"], g["b1"], g["w2"], g["b2"])
    r = K_.predictor_stop(_cuda(g["X"], np.float32), packed, 64, 32, stage_idx=0, L=4, risk_adjustment=False)
    np.testing.assert_allclose(r.score.cpu().numpy(), g["scores"], rtol=0, atol=SCORE_ATOL)
    assert r.k_star is None and r.thr_stop is None


@pytest.mark.parametrize("B,K,V,T", [(32, 8, 30000, 1.0), (64, 8, 20000, 0.7), (40, 32, 5000, 1.0), (8, 8, 30000, 1.0),
                                     (33, 7, 9999, 0.7), (8, 8, 152064, 0.7), (16, 8, 152064, 1.0), (6, 40, 4000, 0.7),
                                     (40, 40, 3000, 1.0), (1, 1, 152064, 0.7)])
def test_in_kernel_epilogue_equals_two_launches(golden, K_, B, K, V, T):
    """asd_verify_accept_fused_ex (N1 second form) == asd_verify_accept_ex followed by asd_predictor_stop, bit for bit,
    on every path of the ONE launch: one workgroup per row (B*K >= CUs: each row hands lp_t over through a self-tagging
    slot), split rows (B*K < CUs, e.g. BASELINE configs[1] B = 8: the finisher wave holds all K lp_t), K > 32 (no
    ballot-by-atomic), with and without the temperature folded in."""
    import torch
    from tests.helpers import make_verify_case, to_device_logits, assert_verify_matches
    g = golden.npz("predictor.npz")
    packed = K_.pack_mlp_weights(g["w1"], g["b1"], g["w2"], g["b2"])
    case = make_verify_case(B, K, V, O.DT_BF16, seed=B + K)
    lg = to_device_logits(case["logits"], case["dtype"]).view(B, K, V)
    tok, lp_d, u = (torch.from_numpy(case[k]).cuda() for k in ("tok", "lp_d", "u"))
    rng = np.random.default_rng(5)
    feat = torch.from_numpy((rng.standard_normal((B, 64)) * 0.3).astype(np.float32)).cuda()
    Cc = torch.tensor([1.0, 4.5, 10.0], dtype=torch.float64, device="cuda")
    theta = torch.tensor([0.6, 0.4, 0.0], dtype=torch.float64, device="cuda")
    ws = K_.VerifyWorkspace(B, K, V)
    inv_t = float(np.float32(1.0 / T))
    for rep in range(3):                                          # repeated calls: the hand-off lines are reusable
        ph1 = torch.ones((B, 3), dtype=torch.float64, device="cuda")
        ph2 = ph1.clone()
        v1 = K_.verify_accept(lg, tok, lp_d, u, ws, inv_temperature=inv_t)
        s1 = K_.predictor_stop(feat, packed, 64, 32, stage_idx=0, L=3, lp=v1.lp_target, stats_col=5, risk_adjustment=True,
                               n_obs=120, alpha=1.0, beta=1.5, p_hist=ph1, Cc=Cc, lam=0.8, theta=theta, want_stats=True)
        v2, s2 = K_.verify_accept_fused(lg, tok, lp_d, u, ws, feat, packed, 64, 32, stage_idx=0, L=3, stats_col=5,
                                        risk_adjustment=True, n_obs=120, alpha=1.0, beta=1.5, p_hist=ph2, Cc=Cc, lam=0.8,
                                        theta=theta, want_stats=True, inv_temperature=inv_t)
        torch.cuda.synchronize()
        assert int(ws.buf.count_nonzero()) == 0                   # every hand-off word is handed back empty
        for a, b in ((v1.lp_target, v2.lp_target), (v1.accept, v2.accept), (v1.n_acc, v2.n_acc),
                     (v1.accept_bits, v2.accept_bits), (s1.stats, s2.stats), (s1.score, s2.score), (s1.k_star, s2.k_star),
                     (s1.stop, s2.stop), (s1.thr_stop, s2.thr_stop), (ph1, ph2)):
            assert torch.equal(a, b)
    got = dict(lp_t=v2.lp_target.cpu().numpy(), accept=v2.accept.cpu().numpy(), n_acc=v2.n_acc.cpu().numpy(),
               bits=v2.accept_bits.cpu().numpy().view(np.uint64))
    if T == 1.0:
        assert_verify_matches(got, case["ref"])
    stats = O.logprob_stats(got["lp_t"], None, K)
    assert s2.stats.cpu().numpy().tobytes() == stats.tobytes()


# ---------------------------------------------------------------------------------------------------------------------
# round 4: the 256 -> 128 -> 1 predictor of the reference's server (QualityPredictor(feature_dim=256), src/serving/server.py:168;
# docs/guides/RESEARCH_PROTOCOL.md:315-364) -- a doc-level spec, no reference arithmetic to pin: parity vs the oracle's generic MLP
def _w256(seed=3):
    rng = np.random.default_rng(seed)
    w1 = (rng.standard_normal((128, 256)) / 16).astype(np.float32)
    b1 = (rng.standard_normal(128) * 0.1).astype(np.float32)
    w2 = (rng.standard_normal((1, 128)) / 8).astype(np.float32)
    b2 = np.array([0.05], np.float32)
    return w1, b1, w2, b2


@pytest.mark.parametrize("B,K", [(1, 8), (33, 8), (256, 16), (40, 40), (8, 3)])
def test_predictor_stop_256x128_latency_form_matches_the_oracle(K_, B, K):
    """asd_predictor_stop with in_dim 256 / hidden 128: one workgroup of eight waves per sequence (k_predictor_stop_w256x128)."""
    import torch
    w1, b1, w2, b2 = _w256()
    packed = K_.pack_mlp_weights(w1, b1, w2, b2)
    rng = np.random.default_rng(B * 7 + K)
    feat = (rng.standard_normal((B, 256)) * 0.3).astype(np.float32)
    lp = (-np.abs(rng.standard_normal((B, K))) * 2).astype(np.float32)
    nv = rng.integers(0, K + 1, B).astype(np.int32)
    Cc = np.array([1.0, 4.5, 10.0])
    for col in (5, 30, 251, -1):
        p_hist = rng.uniform(0.2, 1.0, (B, 3))
        p_hist[:, 2] = 1.0
        ph = _cuda(p_hist, np.float64)
        r = K_.predictor_stop(_cuda(feat, np.float32), packed, 256, 128, stage_idx=1, L=3, lp=_cuda(lp, np.float32),
                              n_valid=_cuda(nv, np.int32), stats_col=col, risk_adjustment=True, n_obs=150, alpha=1.0, beta=2.0,
                              p_hist=ph, Cc=_cuda(Cc, np.float64), lam=0.7, want_stats=True)
        torch.cuda.synchronize()
        stats = O.logprob_stats(lp, nv, K)
        assert r.stats.cpu().numpy().tobytes() == stats.tobytes()
        x = feat.copy()
        if col >= 0:
            x[:, col:col + 5] = stats.astype(np.float32)
        score = O.mlp_predict(x, w1, b1, w2[0], b2)
        got = r.score.cpu().numpy()
        np.testing.assert_allclose(got, score, rtol=0, atol=SCORE_ATOL)
        want_hist = p_hist.copy()
        want_hist[:, 1] = O.bayes_adjust(got.astype(np.float64), 150, 1.0, 2.0)
        assert ph.cpu().numpy().tobytes() == want_hist.tobytes()
        ks, _ = O.optimal_stopping(want_hist, Cc, 0.7)
        assert np.array_equal(r.k_star.cpu().numpy(), ks)


@pytest.mark.parametrize("B,K,V,T", [(32, 8, 30000, 1.0), (64, 4, 20000, 0.7), (16, 16, 9000, 1.0), (32, 8, 152064, 0.7),
                                     (8, 8, 30000, 1.0), (40, 32, 5000, 1.0), (100, 8, 50257, 1.3), (33, 8, 30000, 1.0)])
def test_in_kernel_epilogue_256x128_equals_two_launches(K_, B, K, V, T):
    """asd_verify_accept_fused_ex with the 256 -> 128 -> 1 predictor == asd_verify_accept_ex + asd_predictor_stop, bit for bit:
    in-kernel (k_verify<..., EPI = 2>: the first layer cut over the eight waves of the finisher's workgroup) with one workgroup per
    row and K <= 16; split rows (B * K < CUs, or a row count the geometry heuristic cuts into balanced slices: 100 x 8, 33 x 8) and
    longer drafts take the two-launch route inside the same entry point."""
    import torch
    from tests.helpers import make_verify_case, to_device_logits
    w1, b1, w2, b2 = _w256()
    packed = K_.pack_mlp_weights(w1, b1, w2, b2)
    case = make_verify_case(B, K, V, O.DT_BF16, seed=B + K)
    lg = to_device_logits(case["logits"], case["dtype"]).view(B, K, V)
    tok, lp_d, u = (torch.from_numpy(case[k]).cuda() for k in ("tok", "lp_d", "u"))
    rng = np.random.default_rng(9)
    feat_np = (rng.standard_normal((B, 256)) * 0.3).astype(np.float32)
    feat = torch.from_numpy(feat_np).cuda()
    Cc = torch.tensor([1.0, 4.5, 10.0], dtype=torch.float64, device="cuda")
    theta = torch.tensor([0.6, 0.4, 0.0], dtype=torch.float64, device="cuda")
    ws = K_.VerifyWorkspace(B, K, V)
    inv_t = float(np.float32(1.0 / T))
    for rep in range(3):
        ph1 = torch.ones((B, 3), dtype=torch.float64, device="cuda")
        ph2 = ph1.clone()
        v1 = K_.verify_accept(lg, tok, lp_d, u, ws, inv_temperature=inv_t)
        s1 = K_.predictor_stop(feat, packed, 256, 128, stage_idx=0, L=3, lp=v1.lp_target, stats_col=5, risk_adjustment=True,
                               n_obs=120, alpha=1.0, beta=1.5, p_hist=ph1, Cc=Cc, lam=0.8, theta=theta, want_stats=True)
        v2, s2 = K_.verify_accept_fused(lg, tok, lp_d, u, ws, feat, packed, 256, 128, stage_idx=0, L=3, stats_col=5,
                                        risk_adjustment=True, n_obs=120, alpha=1.0, beta=1.5, p_hist=ph2, Cc=Cc, lam=0.8,
                                        theta=theta, want_stats=True, inv_temperature=inv_t)
        torch.cuda.synchronize()
        assert int(ws.buf.count_nonzero()) == 0
        for a, b in ((v1.lp_target, v2.lp_target), (v1.accept, v2.accept), (v1.n_acc, v2.n_acc), (v1.accept_bits, v2.accept_bits),
                     (s1.stats, s2.stats), (s1.score, s2.score), (s1.k_star, s2.k_star), (s1.stop, s2.stop),
                     (s1.thr_stop, s2.thr_stop), (ph1, ph2)):
            assert torch.equal(a, b)
    stats = O.logprob_stats(v2.lp_target.cpu().numpy(), None, K)
    assert s2.stats.cpu().numpy().tobytes() == stats.tobytes()
    x = feat_np.copy()
    x[:, 5:10] = stats.astype(np.float32)
    np.testing.assert_allclose(s2.score.cpu().numpy(), O.mlp_predict(x, w1, b1, w2[0], b2), rtol=0, atol=SCORE_ATOL)
