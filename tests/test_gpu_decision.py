"""f64 decision kernels (A1, A2, A3, A11) against the golden vectors generated from the
reference's own files: bit-exact.  Called through the C ABI on device tensors."""
import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def K_():
    from asd_amd import kernels
    return kernels


def _cuda(a, dtype):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a, dtype=dtype)).cuda()


def test_optimal_stopping_rule_goldens_bit_exact(golden, K_):
    g = golden.npz("dp_rule.npz")
    N = g["L"].size
    checked = 0
    # group cases that share (L, C, lam, risk, alpha, beta): one batched launch per group
    keys = {}
    for i in range(N):
        L = int(g["L"][i])
        key = (L, g["C"][i, :L].tobytes(), float(g["lam"][i]), int(g["risk"][i]), float(g["alpha"][i]),
               float(g["beta"][i]))
        keys.setdefault(key, []).append(i)
    for (L, cb, lam, risk, al, be), idx in keys.items():
        p = _cuda(g["p"][idx, :L], np.float64)
        Cc = _cuda(np.frombuffer(cb, dtype=np.float64), np.float64)
        k, J = K_.optimal_stopping(p, Cc, lam, bool(risk), al, be)
        assert np.array_equal(k.cpu().numpy(), g["k_star"][idx])
        assert J.cpu().numpy().tobytes() == np.ascontiguousarray(g["J"][idx, :L + 1]).tobytes()
        checked += len(idx)
    assert checked == N >= 1000


def test_large_batch_matches_oracle(K_):
    from oracle import oracle as O
    rng = np.random.default_rng(3)
    for L in (1, 3, 4, 16):
        p = rng.uniform(0, 1, (20000, L))
        p[::7] = np.round(p[::7], 1)
        Cc = np.cumsum(rng.uniform(0.5, 3, L))
        for lam in (0.0, 1.0, 7.5):
            for risk in (False, True):
                k, J = K_.optimal_stopping(_cuda(p, np.float64), _cuda(Cc, np.float64), lam, risk, 1.5, 2.5)
                ko, Jo = O.optimal_stopping(p, Cc, lam, risk, 1.5, 2.5)
                assert np.array_equal(k.cpu().numpy(), ko)
                assert J.cpu().numpy().tobytes() == Jo.tobytes()


def test_length_mismatch_and_limits(K_):
    import torch
    with pytest.raises(ValueError):                     # dp_solver.py:34-35
        K_.optimal_stopping(torch.zeros((2, 3), dtype=torch.float64, device="cuda"),
                            torch.zeros((2,), dtype=torch.float64, device="cuda"), 1.0)
    with pytest.raises(K_.B.AsdError):                  # L > ASD_MAX_STAGES
        K_.optimal_stopping(torch.zeros((2, 17), dtype=torch.float64, device="cuda"),
                            torch.zeros((17,), dtype=torch.float64, device="cuda"), 1.0)


def test_bayes_goldens_bit_exact(golden, K_):
    g = golden.npz("bayes.npz")
    groups = {}
    for i in range(g["p"].size):
        groups.setdefault((int(g["n_obs"][i]), float(g["alpha"][i]), float(g["beta"][i])), []).append(i)
    for (n_obs, al, be), idx in groups.items():
        out = K_.bayes_adjust(_cuda(g["p"][idx], np.float64), n_obs, al, be).cpu().numpy()
        assert out.tobytes() == np.ascontiguousarray(g["out"][idx]).tobytes()


def test_expected_cost_goldens_bit_exact(golden, K_):
    g = golden.npz("dp_rule.npz")
    for i in range(0, g["L"].size, 3):
        L = int(g["L"][i])
        p = _cuda(g["p"][i:i + 1, :L], np.float64)
        Cc = _cuda(g["C"][i, :L], np.float64)
        for kk, want in ((g["k_star"][i], g["cost_at_kstar"][i]), (g["k_rand"][i], g["cost_at_krand"][i])):
            got = K_.expected_cost(p, Cc, float(g["lam"][i]), _cuda([kk], np.int32)).cpu().numpy()
            assert got.tobytes() == np.float64(want).tobytes()


def test_threshold_picks_bit_exact(golden, K_):
    g = golden.npz("predictor.npz")
    picks = golden.json("threshold_picks.json")
    scores = _cuda(g["scores_one_by_one"].astype(np.float32), np.float32)
    for lam, rec in picks.items():
        theta, _ = K_.derive_thresholds([0.7, 0.8, 0.85, 0.9], [1.0, 2.0, 4.5, 10.0], float(lam))
        assert theta.tolist() == rec["theta"]
        got = K_.threshold_stop(scores, _cuda(theta, np.float64)).cpu().numpy()
        assert got.tolist() == rec["stage"]


def test_reference_signature_functions_on_gpu(golden):
    """The list-in / tuple-out API of dp_solver.py, served by the HIP backend."""
    from asd_amd.algorithms import (OptimalStoppingTable, bayesian_adjustment, compute_expected_cost,
                                    optimal_stopping_rule)
    assert optimal_stopping_rule([.3, .5, .8, 1], [1, 1.6, 4.2, 8.8], 100) == (3, [15.6, 14.6, 13.0, 8.8, 0.0])
    assert optimal_stopping_rule([.3, .5, .8, 1], [1, 1.6, 4.2, 8.8], 1.0) == (0, [1.7, 2.45, 5.08, 8.8, 0.0])
    assert bayesian_adjustment(0.25, 100, 1, 1) == 0.2549019607843137
    with pytest.raises(ValueError):
        optimal_stopping_rule([0.5], [1.0, 2.0], 1.0)
    g = golden.json("a4_table_adaptive.json")
    tab = OptimalStoppingTable(g["lambdas"], 4)
    tab.precompute(g["cost"], g["grid"])
    for q in g["lookups"]:
        assert tab.lookup(q["p"], q["lam"]) == q["k"]
        assert tab.lookup(q["p"], q["lam"], fallback_to_dp=False) == q["k_nofallback"]
    d = golden.npz("dp_rule.npz")
    i = 5
    L = int(d["L"][i])
    assert compute_expected_cost(list(d["p"][i, :L]), list(d["C"][i, :L]), float(d["lam"][i]),
                                 int(d["k_star"][i])) == float(d["cost_at_kstar"][i])


def test_lambda_sweep_bit_exact(golden):
    """N4: asd_lambda_sweep -- k*, sum C[:k*+1] and prod p[:k*+1] for every (lambda, request) in one launch --
    against the reference's optimal_stopping_rule / compute_expected_cost outputs (f64, bit for bit)."""
    import torch
    from asd_amd import kernels as K
    g = golden.npz("lambda_sweep.npz")
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
    k, cost, ok = K.lambda_sweep(dev(g["p"]), dev(g["C"]), dev(g["lam"]))
    torch.cuda.synchronize()
    assert np.array_equal(k.cpu().numpy(), g["k_star"])
    assert cost.cpu().numpy().tobytes() == g["cost"].tobytes()
    assert ok.cpu().numpy().tobytes() == g["p_ok"].tobytes()
    # risk-adjusted variant against the oracle (which is pinned by dp_rule.npz)
    k2, c2, o2 = K.lambda_sweep(dev(g["p"]), dev(g["C"]), dev(g["lam"]), risk_adjustment=True, alpha=2.0, beta=3.0)
    rk, rc, ro = O.lambda_sweep(g["p"], g["C"], g["lam"], True, 2.0, 3.0)
    assert np.array_equal(k2.cpu().numpy(), rk) and c2.cpu().numpy().tobytes() == rc.tobytes()
    assert o2.cpu().numpy().tobytes() == ro.tobytes()
    # the controllers on the real backend give the recorded reference results
    import asd_amd
    from asd_amd.algorithms import LambdaOptimizer, StagePopulation
    asd_amd.set_backend(None)
    ref = golden.json("lambda_optimizer.json")
    pop = StagePopulation(g["p"], g["C"], ms_per_cost=ref["ms_per_cost"])
    front = LambdaOptimizer(lambda_bounds=(0.05, 50.0)).optimize_pareto_front(pop.evaluate, 12)
    assert [list(map(float, t)) for t in front] == ref["pareto"]
