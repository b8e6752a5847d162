"""The CPU oracle (oracle/asd_oracle.c + oracle/oracle.py) against the golden vectors generated
from the reference's own files (oracle/gen_golden.py).  Integer / f64 results: bit-exact.
f32 predictor scores: 1e-6.  CPU only."""
import numpy as np
import pytest

from oracle import oracle as O


def test_dp_rule_bit_exact(golden):
    g = golden.npz("dp_rule.npz")
    N = g["L"].size
    for L in (1, 2, 3, 4):
        for risk in (0, 1):
            idx = np.where((g["L"] == L) & (g["risk"] == risk))[0]
            if idx.size == 0:
                continue
            # group by (C, lam, alpha, beta): the batched oracle shares them across a batch
            for i in idx:
                k, J = O.optimal_stopping(g["p"][i, :L], g["C"][i, :L], float(g["lam"][i]),
                                          bool(risk), float(g["alpha"][i]), float(g["beta"][i]))
                assert int(k[0]) == int(g["k_star"][i]), i
                assert J[0].tobytes() == g["J"][i, :L + 1].tobytes(), i
    assert N >= 1000


def test_dp_rule_python_restatement_matches_c(golden):
    g = golden.npz("dp_rule.npz")
    for i in range(0, g["L"].size, 7):
        L = int(g["L"][i])
        k, J = O.py_optimal_stopping_rule(list(map(float, g["p"][i, :L])), list(map(float, g["C"][i, :L])),
                                          float(g["lam"][i]), bool(g["risk"][i]), float(g["alpha"][i]),
                                          float(g["beta"][i]))
        assert k == int(g["k_star"][i])
        assert np.array(J).tobytes() == g["J"][i, :L + 1].tobytes()


def test_dp_rule_length_mismatch_raises():
    with pytest.raises(ValueError):
        O.optimal_stopping([0.5, 0.5], [1.0], 1.0)          # dp_solver.py:34-35
    with pytest.raises(ValueError):
        O.py_optimal_stopping_rule([0.5], [1.0, 2.0], 1.0)


def test_prefix_rule_always_stops_at_stage0():
    """optimal_stopping_rule on an L=1 prefix returns k*=0 for every p, lam."""
    for p0 in (0.0, 0.3, 0.9, 0.999, 1.0):
        for lam in (0.01, 1.0, 100.0):
            k, _ = O.optimal_stopping([p0], [1.0], lam)
            assert int(k[0]) == 0


def test_expected_cost_bit_exact(golden):
    g = golden.npz("dp_rule.npz")
    for i in range(g["L"].size):
        L = int(g["L"][i])
        for kk, want in ((g["k_star"][i], g["cost_at_kstar"][i]), (g["k_rand"][i], g["cost_at_krand"][i])):
            got = O.expected_cost(g["p"][i, :L], g["C"][i, :L], float(g["lam"][i]), [int(kk)])
            assert got.tobytes() == np.float64(want).tobytes(), i


def test_survey_known_answers():
    k, J = O.optimal_stopping([.3, .5, .8, 1], [1, 1.6, 4.2, 8.8], 100.0)
    assert int(k[0]) == 3 and J[0].tolist() == [15.6, 14.6, 13.0, 8.8, 0.0]
    k, J = O.optimal_stopping([.3, .5, .8, 1], [1, 1.6, 4.2, 8.8], 1.0)
    assert int(k[0]) == 0 and J[0].tolist() == [1.7, 2.45, 5.08, 8.8, 0.0]
    assert O.bayes_adjust([0.25], 100, 1, 1)[0] == 0.2549019607843137
    assert O.bayes_adjust([0.9], 1000, 2, 2)[0] == 0.898406374501992
    th, _ = O.derive_thresholds([.7, .8, .85, .9], [1, 2, 4.5, 10], 0.1)
    assert th.tolist() == [0.6363636363636364, 0.48, 0.22580645161290325, 0.0]
    th, _ = O.derive_thresholds([.7, .8, .85, .9], [1, 2, 4.5, 10], 1.0)
    assert th.tolist() == [-0.09999999999999998, -0.4714285714285714, -0.7076923076923076, 0.0]


def test_bayes_bit_exact(golden):
    g = golden.npz("bayes.npz")
    for i in range(g["p"].size):
        got = O.bayes_adjust([g["p"][i]], int(g["n_obs"][i]), float(g["alpha"][i]), float(g["beta"][i]))
        assert got[0].tobytes() == g["out"][i].tobytes(), i
        assert O.py_bayesian_adjustment(float(g["p"][i]), int(g["n_obs"][i]), float(g["alpha"][i]),
                                        float(g["beta"][i])) == float(g["out"][i])


def test_thresholds_bit_exact(golden):
    g = golden.json("thresholds.json")
    for row in g["rows"]:
        th, _ = O.derive_thresholds(row["q"], row["c"], row["lam"])
        assert th.tolist() == row["theta"], row
        assert O.py_derive_optimal_policy(row["q"], row["c"], row["lam"]) == row["theta"]
    th, _ = O.derive_thresholds([0.7, 0.8, 0.85, 0.9], [1.0, 2.0, 4.5, 10.0], 1.0)
    assert th.tolist() == g["misc"]["default_theta"]


def test_predictor_scores(golden):
    g = golden.npz("predictor.npz")
    got = O.mlp_predict(g["X"], g["w1"], g["b1"], g["w2"][0], g["b2"])
    # tolerance: BASELINE.json "stopping scores within 1e-5 fp32"; observed ~1e-7
    np.testing.assert_allclose(got, g["scores"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(got, g["scores_one_by_one"], rtol=0, atol=1e-6)


def test_threshold_picks_bit_exact(golden):
    g = golden.npz("predictor.npz")
    picks = golden.json("threshold_picks.json")
    scores = g["scores_one_by_one"].astype(np.float32)   # .item() of the f32 output
    for lam, rec in picks.items():
        got = O.threshold_stop(scores, rec["theta"])
        assert got.tolist() == rec["stage"], lam


def test_logprob_stats_against_reference_features(golden):
    g = golden.npz("features_a7.npz")
    lp = g["logprobs"].astype(np.float32)                # values are f32-representable by construction
    assert np.array_equal(lp.astype(np.float64), g["logprobs"])
    got = O.logprob_stats(lp, g["n_valid"], K=128)
    want = g["features"][:, 5:10]
    assert got.tobytes() == want.tobytes()
    for i in range(0, lp.shape[0], 9):
        n = int(g["n_valid"][i])
        assert O.py_logprob_stats([float(x) for x in lp[i, :n]]) == want[i].tolist()


def test_token_logprob_matches_reference_idiom(golden):
    """A6: log(softmax(score)[tok]) as the reference computes it (torch f32) vs the f64 oracle."""
    g = golden.npz("logprob_idiom.npz")
    R, V = g["scores"].shape
    res = O.verify_accept(g["scores"], O.DT_F32, g["tok"], np.zeros(R, np.float32),
                          np.full(R, 0.5, np.float32), B=R, K=1, V=V)
    np.testing.assert_allclose(res["lp_t"][:, 0], g["logprob"], rtol=1e-6, atol=1e-5)
    for i in (0, 3, 4, 11):
        assert abs(O.py_token_logprob_reference_idiom(g["scores"][i], int(g["tok"][i])) - g["logprob"][i]) < 1e-5


def test_verify_accept_c_vs_numpy_small():
    rng = np.random.default_rng(7)
    B, K, V = 3, 5, 257
    x = (rng.standard_normal((B, K, V)) * 4).astype(np.float32)
    bits = O.f32_to_bf16_bits(x)
    xf = O.bf16_bits_to_f32(bits).reshape(B, K, V)
    tok = rng.integers(0, V, (B, K)).astype(np.int32)
    tok[0, 1] = -1
    tok[1, 2] = V
    lp_d = -np.abs(rng.standard_normal((B, K))).astype(np.float32)
    u = rng.uniform(0, 1, (B, K)).astype(np.float32)
    u[2, 0] = 0.0
    res = O.verify_accept(bits.reshape(B * K, V), O.DT_BF16, tok, lp_d, u, B, K, V)
    lp, acc, n_acc = O.py_verify_accept(xf, tok, lp_d, u)
    np.testing.assert_allclose(res["lp_t64"], lp, rtol=1e-12, atol=1e-12)
    assert np.array_equal(res["accept"], acc)
    assert np.array_equal(res["n_acc"], n_acc)
    assert res["lp_t"][0, 1] == -np.inf and res["accept"][0, 1] == 0
    for b in range(B):
        w = sum(int(res["accept"][b, k]) << k for k in range(K))
        assert int(res["bits"][b]) == w


def test_lambda_sweep_oracle_matches_reference(golden):
    """N4: the oracle's lambda sweep against the reference's optimal_stopping_rule / compute_expected_cost."""
    g = golden.npz("lambda_sweep.npz")
    k, cost, ok = O.lambda_sweep(g["p"], g["C"], g["lam"])
    assert np.array_equal(k, g["k_star"])
    assert cost.tobytes() == g["cost"].tobytes() and ok.tobytes() == g["p_ok"].tobytes()
    total = cost + g["lam"][:, None] * (1 - ok)                     # compute_expected_cost, dp_solver.py:95-101
    assert total.tobytes() == g["expected_cost"].tobytes()


def test_logprob_idiom_at_full_vocabulary_with_temperature_and_top_p(golden):
    """A6 where the path actually runs: V = 152064, scores that went through bf16 / fp16 storage, divided by
    T = 0.7 and masked to the top-p nucleus by HF's own warpers (generate_training_data.py:110-119), then the
    reference idiom (:128-136) executed as written by oracle/gen_golden.py.  The oracle (f64 LSE) must agree to
    1e-5 whether the temperature is folded into the pass (inv_temperature) or the processed row is given."""
    from tests.helpers import REF_F32_SUM_ERR, encode_logits, full_size_cases
    g = golden.npz("logprob_idiom_full.npz")
    inv_t = np.float32(1.0) / np.float32(g["temperature"])
    n = 0
    worst = {False: 0.0, True: 0.0}
    for var, tok, want, x, keep in full_size_cases(g):
        V = x.size
        one = lambda store, dt, it: O.verify_accept(store.reshape(1, V), dt, [tok], [0.0], [0.5], 1, 1, V,  # noqa: E731
                                                    inv_temperature=it)["lp_t64"][0, 0]
        if var == "f32":
            got = [one(x, O.DT_F32, 1.0)]
        else:
            dt = O.DT_BF16 if var.startswith("bf16") else O.DT_F16
            xs = x.copy()
            if keep is not None:
                mask = np.ones(V, bool)
                mask[keep] = False
                xs[mask] = -np.inf
            processed = (xs / np.float32(g["temperature"])).astype(np.float32)       # the row the reference loop saw
            got = [one(encode_logits(xs, dt), dt, inv_t), one(processed, O.DT_F32, 1.0)]
        for v in got:
            if np.isneginf(want):
                assert np.isneginf(v)
            else:
                # the reference idiom is f32 end to end: at V = 152064 its own softmax sum is 0.3e-5 .. 1.7e-5 below the
                # exact value (measured here against f64: the bias is torch's f32 accumulation, it vanishes on the
                # top-p rows, which have tens of terms) -- so the f64 oracle is held to REF_F32_SUM_ERR, not to 1e-5
                assert abs(v - want) <= REF_F32_SUM_ERR + 1e-6 * abs(want), (var, tok, v, want)
                worst[var.endswith("_topp")] = max(worst[var.endswith("_topp")], abs(v - want))
        assert abs(got[0] - got[-1]) <= 2e-6 or np.isneginf(want)     # folded temperature == processed row
        n += 1
    assert n == 30
    assert worst[True] <= 1e-6, worst         # nucleus rows: the idiom is exact to f32 rounding


def test_draft_sample_oracle_reproduces_the_hf_warpers_nucleus(golden):
    """X1: the oracle's top-p rule against transformers' TemperatureLogitsWarper + TopPLogitsWarper (what the reference's
    generate(temperature=0.7, top_p=0.9) call applies, generate_training_data.py:110-119) on 54 rows: three vocabulary
    sizes, three storage types, wide / tight / flat rows."""
    from helpers import check_nucleus_against_warper, nucleus_cases
    g = golden.npz("top_p_nucleus.npz")
    n = 0
    for c in nucleus_cases(g):
        inv_t = float(np.float32(1.0) / np.float32(c["T"]))
        r = np.array([0.37], np.float32)
        ref = O.draft_sample(c["store"], c["dtype"], r, 1, c["V"], inv_t, c["top_p"])
        check_nucleus_against_warper(c, int(ref["tok"][0]), float(ref["lp"][0]), ref["thr"][0], 2e-6)
        n += 1
    assert n == 54


def test_accept_rule_and_residual_distribution_against_hf_speculative_sampling(golden):
    """A5 + the residual draw: the oracle against transformers' `_speculative_sampling` (assisted generation; algorithm 1 of
    the speculative-decoding paper) called unmodified with its uniforms supplied from outside: same number of accepted
    drafts on all 24 cases (n_matches 0 .. K), and the inverse CDF of HF's residual distribution p' = norm(max(0, p - q))
    (or of p_{n+1} after a fully accepted block) picks the token the oracle picks wherever the draw is >= 1e-5 of the
    mass away from a CDF edge."""
    from helpers import spec_cases
    g = golden.npz("speculative_sampling.npz")
    n_cases = n_draws = 0
    for c in spec_cases(g):
        K, V = c["K"], c["V"]
        ref = O.verify_accept(c["new"][:K], O.DT_F32, c["tok"], c["lp_d"], c["u"], 1, K, V)
        assert int(ref["n_acc"][0]) == c["n_matches"]
        for r, want, margin in zip(c["r"], c["want_tok"], c["margin"]):
            tok, _ = O.residual_sample(c["new"][:K], c["cand"], O.DT_F32, [c["n_matches"]], [r], 1, K, V, bonus=c["new"][K:K + 1])
            if margin > 1e-5:
                assert int(tok[0]) == int(want)
                n_draws += 1
        n_cases += 1
    assert n_cases == 24 and n_draws >= 60


def test_accept_residual_and_proposal_at_the_full_vocabulary_against_hf(golden):
    """The same pin at the size and settings the path RUNS at: V = 152064, rows stored as bf16 / f16,
    the reference's T = 0.7 / top_p = 0.9 (generate_training_data.py:110-119) -- the draft scores warped by HF's
    TemperatureLogitsWarper + TopPLogitsWarper, `_speculative_sampling` called unmodified
    (oracle/gen_golden.py::gen_speculative_sampling_full).  Oracle: n_acc == HF's n_matches on all 16 cases; the residual
    draw against the NUCLEUS-TRUNCATED draft row (x* thresholds) picks HF's token wherever the draw is >= 1e-5 of the mass
    from a CDF edge; the proposal step reproduces HF's nucleus threshold, drafted token and log q(token)."""
    from helpers import spec_full_cases
    g = golden.npz("speculative_sampling_full.npz")
    n_cases = n_draws = n_prop = 0
    for c in spec_full_cases(g):
        K, V, dt = c["K"], c["V"], c["dtype"]
        ref = O.verify_accept(c["new"][:K], dt, c["tok"], c["lq"].astype(np.float32), c["u"], 1, K, V, inv_temperature=c["inv_t"])
        assert int(ref["n_acc"][0]) == c["n_matches"], c["case"]
        # HF's warper drops some of the scores EQUAL to the nucleus threshold (its sort order decides), the kernels keep every
        # tie: where that happened on the row the residual is taken against, q -- and with it p' -- differs by ~1e-3 of mass
        same_residual = c["n_matches"] == K or c["ties_removed"][c["n_matches"]] == 0
        for r, want, margin in zip(c["r"], c["want_tok"], c["margin"]):
            tok, _ = O.residual_sample(c["new"][:K], c["cand"], dt, [c["n_matches"]], [r], 1, K, V, bonus=c["new"][K:K + 1],
                                       inv_temperature=c["inv_t"], d_threshold=c["thr"].reshape(1, K))
            if margin > 1e-5 and same_residual:
                assert int(tok[0]) == int(want), c["case"]
                n_draws += 1
        d = O.draft_sample(c["cand"], dt, c["pick"], K, V, c["inv_t"], c["top_p"])
        ok = d["margin_p"] > 1e-5
        assert np.array_equal(d["thr"][ok], c["thr"][ok]), c["case"]
        okt = ok & (c["pick_margin"] > 1e-5) & (d["margin_r"] > 1e-5) & (c["ties_removed"] == 0)   # same nucleus as HF's
        assert np.array_equal(d["tok"][okt], c["tok"][okt]), c["case"]
        np.testing.assert_allclose(d["lp"][okt], c["lq"][okt], rtol=0, atol=2e-5)
        n_prop += int(okt.sum())
        n_cases += 1
    assert n_cases == 16 and n_draws >= 24 and n_prop >= 24
