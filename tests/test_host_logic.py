"""Host-side logic of the package (reference-API mirrors) on CPU, with the oracle injected as the
backend (tests/oracle_backend.py).  What is checked here is control flow, signatures, error
behaviour and the pure-host arithmetic (A9, A12, closed forms) against the goldens; the GPU
arithmetic itself is covered by the -m gpu tests."""
import asyncio
import os

import numpy as np
import pytest
import yaml

import asd_amd
from asd_amd.algorithms import (AdaptiveStopping, OptimalStoppingTable, bayesian_adjustment, compute_expected_cost,
                                optimal_stopping_rule, optimal_stopping_rule_batch)
from asd_amd.minimal_adaptive_decoder import (DecodingResult, MinimalAdaptiveDecoder, MinimalQualityPredictor,
                                              SimpleTokenizer, features_from_token_ids, train_minimal_predictor)
from asd_amd.serving import AdaptiveSpeculativePipeline, PipelineConfig, RequestCache, RequestResult
from asd_amd.theory import OptimalStoppingTheory, RegretAnalyzer, TheoreticalParameters
from asd_amd.training import extract_features, extract_features_batch, token_logprobs
from oracle import oracle as O
from tests.oracle_backend import OracleBackend


@pytest.fixture(autouse=True)
def oracle_backend():
    asd_amd.set_backend(OracleBackend())
    yield
    asd_amd.set_backend(None)


# ------------------------------------------------------------------------- dp_solver API
def test_dp_solver_signatures_and_errors(golden):
    k, J = optimal_stopping_rule([.3, .5, .8, 1], [1, 1.6, 4.2, 8.8], 100)
    assert (k, J) == (3, [15.6, 14.6, 13.0, 8.8, 0.0]) and isinstance(k, int) and isinstance(J, list)
    with pytest.raises(ValueError, match="same length"):
        optimal_stopping_rule([0.5, 0.5], [1.0], 1.0)
    with pytest.raises(ValueError):
        optimal_stopping_rule_batch([[0.5, 0.5]], [1.0], 1.0)
    assert bayesian_adjustment(0.9, 1000, 2, 2) == 0.898406374501992
    g = golden.npz("dp_rule.npz")
    for i in (0, 17, 400, 1599):
        L = int(g["L"][i])
        p, c = [float(x) for x in g["p"][i, :L]], [float(x) for x in g["C"][i, :L]]
        k, J = optimal_stopping_rule(p, c, float(g["lam"][i]), bool(g["risk"][i]), float(g["alpha"][i]),
                                     float(g["beta"][i]))
        assert k == int(g["k_star"][i]) and J == g["J"][i, :L + 1].tolist()
        assert compute_expected_cost(p, c, float(g["lam"][i]), int(g["k_star"][i])) == float(g["cost_at_kstar"][i])


def test_table_and_adaptive_stopping_goldens(golden):
    g = golden.json("a4_table_adaptive.json")
    tab = OptimalStoppingTable(g["lambdas"], 4)
    tab.precompute(g["cost"], g["grid"])
    assert set(tab.table) == set(g["lambdas"])
    for q in g["lookups"]:
        assert tab.lookup(q["p"], q["lam"]) == q["k"]
        assert tab.lookup(q["p"], q["lam"], fallback_to_dp=False) == q["k_nofallback"]
    ad = AdaptiveStopping(initial_lambda=0.7, confidence_level=0.1)
    for st, q, lat in g["updates"]:
        ad.update_statistics(int(st), q, lat)
    assert ad.stage_counts.tolist() == g["counts"] and ad.stage_rewards.tolist() == g["rewards"]
    assert [[float(x) for x in ad.get_confidence_bounds(s)] for s in range(4)] == g["bounds"]
    assert [ad.should_explore(s) for s in range(4)] == g["explore"]
    fresh = AdaptiveStopping()
    assert [str(x) for x in fresh.get_confidence_bounds(0)] == g["fresh_bounds"]
    assert fresh.should_explore(2) == g["fresh_explore"]


# ------------------------------------------------------------------------- theory API
def test_theory_thresholds_and_closed_forms(golden):
    g = golden.json("thresholds.json")
    for row in g["rows"]:
        t = OptimalStoppingTheory(TheoreticalParameters(n_stages=len(row["q"]), quality_bounds=list(row["q"]),
                                                        cost_ratios=list(row["c"]), lambda_param=row["lam"]))
        pol = t.derive_optimal_policy()
        assert [float(pol[s]) for s in range(len(row["q"]))] == row["theta"]
        assert list(pol) == list(range(len(row["q"]) - 1, -1, -1))       # reference dict order
        assert t.thresholds_array().tolist() == row["theta"]
    m = g["misc"]
    t = OptimalStoppingTheory(TheoreticalParameters(lambda_param=1.0))
    assert t.params.quality_bounds == [0.7, 0.8, 0.85, 0.9] and t.params.cost_ratios == [1.0, 2.0, 4.5, 10.0]
    assert [float(t.derive_optimal_policy()[s]) for s in range(4)] == m["default_theta"]
    for T, want in m["regret_bound"].items():
        assert float(t.compute_regret_bound(int(T))) == want
    assert t.sample_complexity() == m["sample_complexity"]
    assert t._compute_improvement_probability(1) == 0.6 * (1 - 0.8)
    ra = RegretAnalyzer(t)
    for s, d, want in m["instant_regret"]:
        assert float(ra.compute_instantaneous_regret(s, d)) == want
    assert float(ra.compute_cumulative_regret()) == m["cumulative"]
    assert float(ra.compute_average_regret()) == m["average"]
    assert {k: float(v) for k, v in ra.theoretical_vs_empirical().items()} == m["tve"]
    assert RegretAnalyzer(t).compute_average_regret() == 0.0


# ------------------------------------------------------------------------- minimal decoder
@pytest.fixture
def decoder(tmp_path, golden):
    cfg = golden.json("decoder_misc.json")["config"]
    for i, s in enumerate(cfg["models"]["stages"]):
        s["model_path"] = f"local/stage{i}"
    path = tmp_path / "models.yaml"
    path.write_text(yaml.safe_dump(cfg))
    g = golden.npz("predictor.npz")
    import torch
    pred = MinimalQualityPredictor()
    pred.load_state_dict({"net.0.weight": torch.from_numpy(g["w1"]), "net.0.bias": torch.from_numpy(g["b1"]),
                          "net.3.weight": torch.from_numpy(g["w2"]), "net.3.bias": torch.from_numpy(g["b2"])})
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        yield MinimalAdaptiveDecoder(str(path), predictor=pred)
    finally:
        os.chdir(cwd)


def test_decoder_features_a9_and_heuristics_a12(golden, decoder):
    g = golden.json("decoder_misc.json")

    class Tok:
        def __init__(self, ids):
            self.ids = ids

        def encode(self, prompt, return_tensors="pt"):
            import torch
            return torch.tensor([self.ids], dtype=torch.int64)

    for rec in g["a9"]:
        f = decoder.predictor.extract_features(rec["prompt"], Tok(rec["ids"]))
        assert f.dtype.is_floating_point and f.shape == (64,)
        assert f.tolist() == rec["features"]
        assert features_from_token_ids(rec["ids"], rec["prompt"]).tolist() == rec["features"]
    for rec in g["a12"]:
        d = decoder._estimate_difficulty(rec["prompt"])
        assert d == rec["difficulty"]
        assert [float(decoder._compute_regret(s, d)) for s in range(4)] == rec["regret"]


def test_decoder_decode_matches_golden_picks(golden, decoder):
    g = golden.npz("predictor.npz")
    picks = golden.json("threshold_picks.json")
    assert len(decoder.models) == 4 and isinstance(decoder.tokenizer, SimpleTokenizer)
    assert decoder.predictor.training is False
    scores = decoder._scores(g["X"])
    np.testing.assert_allclose(scores, g["scores"], atol=1e-6, rtol=0)
    for lam, rec in picks.items():
        decoder.set_lambda(float(lam))
        assert decoder._theta_vector().tolist() == rec["theta"]
        got = asd_amd.get_backend().threshold_stop(g["scores_one_by_one"].astype(np.float32), decoder._theta_vector())
        assert got.tolist() == rec["stage"]
    decoder.set_lambda(0.1)
    res = decoder.decode("What is the capital of France?", max_tokens=10)
    assert isinstance(res, DecodingResult) and 0 <= res.selected_stage < 4
    assert res.text == f"[Generated with Qwen3-{['7b', '14b', '32b', '72b'][res.selected_stage]}]"
    batch = decoder.decode_batch(["hi", "why is the sky blue?", "x " * 300])
    assert [r.selected_stage for r in batch] == [decoder.decode(p).selected_stage for p in ["hi", "why is the sky blue?", "x " * 300]]
    assert decoder.decode_batch([]) == []


def test_predictor_module_eval_train_and_training_loop(tmp_path, golden):
    import torch
    g = golden.npz("predictor.npz")
    pred = MinimalQualityPredictor()
    assert sorted(pred.state_dict()) == ["net.0.bias", "net.0.weight", "net.3.bias", "net.3.weight"]
    pred.load_state_dict({"net.0.weight": torch.from_numpy(g["w1"]), "net.0.bias": torch.from_numpy(g["b1"]),
                          "net.3.weight": torch.from_numpy(g["w2"]), "net.3.bias": torch.from_numpy(g["b2"])})
    pred.eval()
    out = pred(torch.from_numpy(g["X"]))
    assert out.shape == (256, 1)
    np.testing.assert_allclose(out.numpy()[:, 0], g["scores"], atol=1e-6, rtol=0)
    assert pred(torch.from_numpy(g["X"][3])).shape == (1,)
    data = [{"features": torch.rand(8, 64), "quality_labels": torch.rand(8, 1).round()} for _ in range(3)]
    trained = train_minimal_predictor(data, data[:1], epochs=2, save_path=str(tmp_path / "ck" / "p.pt"))
    assert (tmp_path / "ck" / "p.pt").exists() and trained.training is False


# ------------------------------------------------------------------------- A6 / A7
def test_extract_features_matches_reference_goldens(golden):
    g = golden.npz("features_a7.npz")
    meta = golden.json("features_a7_meta.json")
    mds = []
    for i, m in enumerate(meta):
        n = int(g["n_valid"][i])
        mds.append({"logprobs": [float(x) for x in g["logprobs"][i, :n]], "generation_time": m["generation_time"],
                    "completion_tokens": m["completion_tokens"]})
    got = extract_features_batch([m["prompt"] for m in meta], [m["output"] for m in meta], mds,
                                 [m["stage_id"] for m in meta])
    assert got.tobytes() == g["features"].tobytes()
    one = extract_features(meta[7]["prompt"], meta[7]["output"], mds[7], meta[7]["stage_id"])
    assert one == g["features"][7].tolist() and len(one) == 64


def test_token_logprobs_matches_reference_idiom(golden):
    g = golden.npz("logprob_idiom.npz")
    lp = token_logprobs(g["scores"], g["tok"])
    np.testing.assert_allclose(lp, g["logprob"], rtol=1e-6, atol=1e-5)


# ------------------------------------------------------------------------- pipeline
class FakeStage:
    def __init__(self, name, cost):
        self.name, self.cost_per_token, self.calls = name, cost, []

    def generate(self, prompts, max_tokens, temperature, return_logprobs=True):
        self.calls.append(list(prompts))
        texts = [f"{self.name} answer to <{p[:12]}>" for p in prompts]
        return texts, [np.array([-0.1, -0.2, -0.3]) for _ in prompts], {"generation_time_ms": 1.0}

    def get_model_info(self):
        return {"name": self.name}


class FakeStageManager:
    def __init__(self, costs=(1.0, 1.6, 4.2, 8.8), names=("8b", "13b", "34b", "70b")):
        self.stages = {n: FakeStage(n, c) for n, c in zip(names, costs)}

    def get_stage(self, name):
        return self.stages[name]


class ScriptedPredictor:
    """p depends on the prompt's first word and the stage: 'easy' -> high, 'hard' -> low."""

    def __init__(self):
        self.calls = 0

    def predict(self, prompt, draft_output, draft_logprobs, stage_id, feature_extractor):
        self.calls += 1
        base = {"easy": 0.97, "mid": 0.6, "hard": 0.05}.get(prompt.split()[0], 0.5)
        return min(0.99, base + 0.2 * stage_id)


def _pipeline(stop_rule, lam=1.0, **kw):
    sm = FakeStageManager()
    cfg = PipelineConfig(lambda_value=lam, stop_rule=stop_rule, **kw)
    return AdaptiveSpeculativePipeline(sm, ScriptedPredictor(), object(), cfg), sm


def test_prefix_rule_reproduces_reference_trace():
    """with the reference's prefix DP every request stops at stage 0 with
    stage_probabilities = [bayes(p0)] and stage_costs = [C0]."""
    pipe, sm = _pipeline("prefix", lam=50.0)
    for word in ("easy", "mid", "hard"):
        r = pipe.process_request(f"{word} question")
        assert isinstance(r, RequestResult)
        assert r.stopped_at_stage == 0 and r.stage_costs == [1.0]
        p0 = {"easy": 0.97, "mid": 0.6, "hard": 0.05}[word]
        n_obs = max(100, pipe.stats["total_requests"] - 1)
        assert r.stage_probabilities == [O.py_bayesian_adjustment(p0, n_obs, 1.0, 1.0)]
        assert r.output.startswith("8b answer")
    assert len(sm.stages["13b"].calls) == 0
    pipe.shutdown()


def test_full_rule_follows_the_dp_over_all_stages():
    pipe, sm = _pipeline("full", lam=30.0, risk_adjustment=False, batch_grouping="none")
    res = pipe.batch_process(["easy one", "hard one", "mid one", "hard two"])
    costs = [1.0, 1.6, 4.2, 8.8]
    for r, word in zip(res, ("easy", "hard", "mid", "hard")):
        base = {"easy": 0.97, "mid": 0.6, "hard": 0.05}[word]
        probs, stop = [], None
        for i in range(4):
            probs.append(1.0 if i == 3 else min(0.99, base + 0.2 * i))
            P = [1.0] * 4
            P[:i + 1] = probs
            k, _ = O.py_optimal_stopping_rule(P, costs, 30.0)
            if k <= i or i == 3:
                stop = k
                break
        assert r.stopped_at_stage == stop
        assert r.stage_probabilities == probs and r.stage_costs == costs[:stop + 1]
        assert r.total_tokens > 0 and r.latency_ms >= 0
        assert r.stages_run == len(probs) >= stop + 1 and r.executed_costs == costs[:len(probs)]
    # batching: one generate() call per stage, shrinking as requests stop
    assert [len(c) for c in sm.stages["8b"].calls] == [4]
    assert all(len(c) <= 4 for c in sm.stages["13b"].calls) and len(sm.stages["13b"].calls) <= 1
    st = pipe.get_stats()
    assert st["total_requests"] == 4 and sum(st["stage_stops"]) == 4
    for key in ("stage_distribution", "avg_tokens_per_request", "cache_stats", "active_requests", "avg_latency",
                "avg_tokens_per_second", "total_tokens", "avg_stage_probabilities", "error_count"):
        assert key in st
    pipe.update_lambda(0.01)
    assert pipe.config.lambda_value == 0.01
    assert pipe.process_request("hard again").stopped_at_stage == 0       # quality is cheap now
    pipe.reset_stats()
    assert pipe.get_stats()["total_requests"] == 0
    pipe.warmup(3)
    assert pipe.get_stats()["total_requests"] == 3
    r = asyncio.run(pipe.process_request_async("easy async", request_id="abc"))
    assert r.request_id == "abc"
    pipe.shutdown()


def test_batch_process_groups_requests_by_predicted_stop_stage():
    """The reference's TODO (pipeline.py:331-338): requests are grouped by the stage ONE DP launch predicts from
    the prompt-only scores, shallow groups run first; per request the result is what the ungrouped batch gives."""
    prompts = ["easy a", "hard a", "mid a", "hard b", "easy b", "mid b", "hard c"]
    plain, _ = _pipeline("full", lam=30.0, risk_adjustment=False, batch_grouping="none")
    want = plain.batch_process(prompts)
    assert plain.config.batch_grouping == "none" and _pipeline("full")[0].config.batch_grouping == "none"   # grouping is opt-in
    pipe, sm = _pipeline("full", lam=30.0, risk_adjustment=False, batch_grouping="predicted_stage")
    pred = pipe.predict_stop_stages(prompts)
    costs = [1.0, 1.6, 4.2, 8.8]
    for word, k in zip((p.split()[0] for p in prompts), pred):
        base = {"easy": 0.97, "mid": 0.6, "hard": 0.05}[word]
        P = [min(0.99, base + 0.2 * i) for i in range(3)] + [1.0]
        assert k == O.py_optimal_stopping_rule(P, costs, 30.0)[0]
    assert len(set(pred.tolist())) >= 2                                  # the population really splits
    got = pipe.batch_process(prompts)
    for a, b, k in zip(got, want, pred):
        assert (a.output, a.stopped_at_stage, a.stage_probabilities, a.stage_costs, a.stages_run) == \
               (b.output, b.stopped_at_stage, b.stage_probabilities, b.stage_costs, b.stages_run)
        assert a.predicted_stage == k and b.predicted_stage == -1
    # one generate() call per (group, stage) and every group enters stage 0 whole
    sizes = sorted(len(c) for c in sm.stages["8b"].calls)
    assert sizes == sorted(int((pred == s).sum()) for s in set(pred.tolist()))
    assert [len(c) for c in sm.stages["8b"].calls] == [int((pred == s).sum()) for s in sorted(set(pred.tolist()))]
    # a predictor with a batch interface is asked ONCE per stage for the whole batch
    class BatchedPredictor(ScriptedPredictor):
        def __init__(self):
            super().__init__()
            self.batch_calls = 0

        def predict_batch(self, prompts, draft_outputs, draft_logprobs, stage_id, feature_extractor):
            self.batch_calls += 1
            return [min(0.99, {"easy": 0.97, "mid": 0.6, "hard": 0.05}.get(p.split()[0], 0.5) + 0.2 * stage_id) for p in prompts]
    bp = BatchedPredictor()
    both = AdaptiveSpeculativePipeline(FakeStageManager(), bp, object(),
                                       PipelineConfig(lambda_value=30.0, stop_rule="full", risk_adjustment=False,
                                                      batch_grouping="predicted_stage"))
    assert np.array_equal(both.predict_stop_stages(prompts), pred) and bp.batch_calls == 3 and bp.calls == 0
    both.shutdown()
    plain.shutdown()
    pipe.shutdown()


def test_dynamic_lambda_rule_matches_the_reference_and_drives_the_pipeline(golden):
    """optimize_lambda_parameter == DynamicCostOptimizer._optimize_lambda_parameter (dynamic_cost_optimizer.py:425-487)
    bit for bit on 240 reference-generated cases; DynamicLambdaController feeds it from pipeline.get_stats()."""
    from asd_amd.algorithms.optimizer import DynamicLambdaController, optimize_lambda_parameter
    for c in golden.json("dynamic_lambda.json"):
        got = optimize_lambda_parameter(c["current_lambda"], c["metrics"], c["gpu_utilization"], c["request_rate"],
                                        c["load_forecast"])
        assert got == c["new_lambda"], c
    pipe, sm = _pipeline("full", lam=1.0, risk_adjustment=False)
    pipe.batch_process(["easy a", "hard a", "mid a"])
    ctl = DynamicLambdaController(pipe, [1.0, 1.6, 4.2, 8.8], target_latency=1e9, min_quality=0.0)
    m = ctl.metrics()
    assert set(m) == {"avg_latency", "avg_quality", "avg_cost"} and m["avg_cost"] >= 1.0
    new = ctl.step(gpu_utilization=[0.5, 0.5])
    # latency far below target (+0.05), quality above min, cost >= 1 ... : whatever the sum, the pipeline follows it
    assert pipe.config.lambda_value == new == optimize_lambda_parameter(1.0, m, [0.5, 0.5], 0.0, [], 1e9, 0.0)
    assert ctl.history == [(1.0, new)]
    pipe.shutdown()


def test_pipeline_errors_are_counted_and_reraised():
    pipe, sm = _pipeline("full")

    def boom(**kw):
        raise RuntimeError("stage down")

    sm.stages["8b"].generate = boom
    with pytest.raises(RuntimeError, match="stage down"):
        pipe.process_request("easy x")
    assert pipe.stats["error_count"] == 1 and pipe.active_requests == {}
    with pytest.raises(ValueError):
        AdaptiveSpeculativePipeline(sm, ScriptedPredictor(), None, PipelineConfig(stop_rule="nope"))
    pipe.shutdown()


def test_request_cache_surface_and_cache_hits():
    c = RequestCache(max_entries=3)
    assert c.get_cache("r", 0) is None
    assert c.allocate("r", 0, {"output": "a", "logprobs": np.array([])})
    c.allocate("r", 1, {"output": "b"})
    c.allocate("r", 2, {"output": "c"})
    assert c.get_cache("r", 1)["output"] == "b"
    c.truncate_at_stage("r", 0)
    assert c.get_cache("r", 1) is None and c.get_cache("r", 0)["output"] == "a"
    c.allocate("q", 0, {"output": "z"})
    c.allocate("q", 1, {"output": "z"})
    c.allocate("s", 0, {"output": "evicts the oldest request"})
    st = c.get_stats()
    assert st["total_allocations"] == 6 and st["evictions"] >= 1
    c.cleanup_request("q")
    # a pre-populated cache entry is used instead of generate()
    pipe, sm = _pipeline("prefix")
    pipe.cache_manager.allocate("rid", 0, {"output": "cached text", "logprobs": np.array([])})
    r = pipe.process_request("easy q", request_id="rid")
    assert r.cache_hits == 1 and r.output == "cached text" and sm.stages["8b"].calls == []
    pipe.shutdown()


def test_doc_spec_predictor_and_feature_extractor():
    """A14 (RESEARCH_PROTOCOL.md:315-409): doc-only spec, so the check is self-consistency -- the kernel-served
    eval forward equals the plain torch module, features follow the documented formulas."""
    import torch
    from asd_amd.serving import FeatureExtractor, QualityPredictor
    torch.manual_seed(3)
    qp = QualityPredictor()
    assert sorted(qp.state_dict()) == ["mlp.0.bias", "mlp.0.weight", "mlp.3.bias", "mlp.3.weight"]
    fx = FeatureExtractor()
    lps = np.log(np.array([[0.5, 0.2, 0.1, 0.1, 0.1], [0.9, 0.05, 0.03, 0.01, 0.01]]))
    f = fx.extract("a b c", "x y", lps, 2)
    assert f.shape == (256,) and f[1] == 3 / 2048 and f[2] == 2 / 512 and f[4] == 0.5
    assert abs(f[0] - (-np.mean([np.sum(np.exp(lp) * lp) for lp in lps]))) < 1e-15
    assert f[3] == np.mean([lp.max() for lp in lps]) and np.all(f[5:] == 0)
    assert fx.extract("a", "b", None, 0)[3] == -10.0 and fx.extract("a", "b", [], 0)[0] == 0.0
    x = torch.randn(7, 256)
    with torch.no_grad():
        want = qp.mlp(x).numpy()
    np.testing.assert_allclose(qp(x).numpy(), want, atol=1e-5, rtol=0)
    p = qp.predict(prompt="a b c", draft_output="x y", draft_logprobs=lps, stage_id=2, feature_extractor=fx)
    with torch.no_grad():
        ref = qp.mlp(torch.from_numpy(f.astype(np.float32))[None]).item()
    assert abs(p - ref) < 1e-5
    # the pipeline accepts it as its predictor
    pipe = AdaptiveSpeculativePipeline(FakeStageManager(), qp, fx, PipelineConfig(stop_rule="full", lambda_value=5.0))
    r = pipe.process_request("easy question")
    assert 0 <= r.stopped_at_stage < 4
    pipe.shutdown()


def test_config_surface_yaml(tmp_path):
    """serving.yaml -> PipelineConfig; the shipped 3-tier models.yaml and the reference's own stage schema
    (without theoretical_quality / relative_cost) both build a decoder."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = PipelineConfig.from_yaml(os.path.join(root, "configs", "serving.yaml"))
    assert cfg.lambda_value == 1.0 and cfg.risk_adjustment is True and cfg.risk_alpha == 1.0 and cfg.risk_beta == 1.0
    assert cfg.batch_timeout_ms == 50.0 and cfg.stop_rule == "full" and cfg.stage_names == ("7b", "32b", "72b")
    (tmp_path / "s.yaml").write_text("pipeline:\n  lambda_value: 2.5\n  risk_adjustment:\n    enabled: false\n    alpha: 2.0\n")
    c2 = PipelineConfig.from_yaml(str(tmp_path / "s.yaml"))
    assert (c2.lambda_value, c2.risk_adjustment, c2.risk_alpha, c2.stop_rule) == (2.5, False, 2.0, "full")
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        dec = MinimalAdaptiveDecoder(os.path.join(root, "configs", "models.yaml"))
        assert len(dec.models) == 3 and dec.theory.params.cost_ratios == [1.0, 4.5, 10.0]
        assert dec._theta_vector().tolist() == O.derive_thresholds([0.7, 0.85, 0.9], [1.0, 4.5, 10.0], 1.0)[0].tolist()
        ref_style = {"models": {"stages": [{"name": "qwen3-7b", "model_path": "Qwen/Qwen3-7B-Instruct", "size_label": "7b"},
                                           {"name": "qwen3-14b", "model_path": "x", "size_label": "14b"},
                                           {"name": "qwen3-32b", "model_path": "x", "size_label": "32b"},
                                           {"name": "qwen3-72b", "model_path": "x", "size_label": "72b"}]}}
        (tmp_path / "ref.yaml").write_text(yaml.safe_dump(ref_style))
        dec2 = MinimalAdaptiveDecoder(str(tmp_path / "ref.yaml"))
        assert dec2.theory.params.quality_bounds == [0.7, 0.8, 0.85, 0.9]
        assert dec2.theory.params.cost_ratios == [1.0, 2.0, 4.5, 10.0]
        assert 0 <= dec2.decode("why?").selected_stage < 4
    finally:
        os.chdir(cwd)


def test_ragged_kv_forward_equals_full_context_and_survives_rollback():
    """N3 plumbing (CPU, fp32): per-sequence positions reproduce the dense forward; after a 'rollback'
    (a shorter length on the caller's side) stale KV entries are overwritten before they are read."""
    import torch

    from asd_amd.serving import synthetic_lm as SL

    lm = SL.SyntheticLM(SL.tiny(vocab=60, hidden=32, layers=2, heads=4, kv_heads=2), dtype=torch.float32, device="cpu", seed=1)
    g = torch.Generator().manual_seed(0)
    ids = torch.randint(0, 60, (3, 9), generator=g)
    full = lm(ids)
    lm.reset()
    lm.alloc_ragged(3, 16)
    zero = torch.zeros(3, dtype=torch.int64)
    a = lm.forward_ragged(ids[:, :5], zero, 12)
    # garbage continuation at per-sequence positions (a rejected draft), then the real tokens over it
    lm.forward_ragged(torch.randint(0, 60, (3, 3), generator=g), torch.tensor([5, 4, 3]), 12)
    # sequence b resumes from its own length (5, 4, 3) with the true tokens
    pos0 = torch.tensor([5, 4, 3])
    T = 4
    chunk = torch.stack([ids[b, pos0[b]:pos0[b] + T] for b in range(3)])
    out = lm.forward_ragged(chunk, pos0, 12)
    for b in range(3):
        lo = int(pos0[b])
        torch.testing.assert_close(out[b], full[b, lo:lo + T], rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(a, full[:, :5], rtol=1e-4, atol=1e-5)


def test_lambda_controllers_match_reference_results(golden, oracle_backend):
    """N4: LambdaOptimizer on the same evaluate functions as the reference run recorded in the goldens."""
    from asd_amd.algorithms import LambdaOptimizer, StagePopulation

    ref = golden.json("lambda_optimizer.json")
    g = golden.npz("lambda_sweep.npz")
    pop = StagePopulation(g["p"], g["C"], ms_per_cost=ref["ms_per_cost"])
    for row in ref["latency"]:
        r = LambdaOptimizer(latency_constraint=row["constraint"]).optimize_for_latency_constraint(pop.evaluate)
        assert (r.optimal_lambda, r.achieved_latency, r.achieved_quality, r.constraint_satisfied, r.iterations) == \
            (row["optimal_lambda"], row["achieved_latency"], row["achieved_quality"], row["constraint_satisfied"], row["iterations"])
    front = LambdaOptimizer(lambda_bounds=(0.05, 50.0)).optimize_pareto_front(pop.evaluate, 12)
    assert [list(map(float, t)) for t in front] == ref["pareto"]
    for row in ref["balanced"]:
        r = LambdaOptimizer().find_balanced_lambda(pop.evaluate, quality_weight=row["quality_weight"])
        assert abs(r.optimal_lambda - row["optimal_lambda"]) <= 1e-9 * max(1.0, row["optimal_lambda"])
        assert (r.achieved_latency, r.achieved_quality, r.iterations) == (row["achieved_latency"], row["achieved_quality"], row["iterations"])

    def analytic(lam):
        return 2000.0 / (1.0 + lam) + 40.0, 1.0 / (1.0 + 0.3 * lam)
    a = ref["analytic"]
    for row in a["latency"]:
        r = LambdaOptimizer(latency_constraint=row["constraint"]).optimize_for_latency_constraint(
            analytic, tolerance=row["tolerance"], max_iterations=row["max_iterations"])
        assert (r.optimal_lambda, r.achieved_latency, r.achieved_quality, r.constraint_satisfied, r.iterations) == \
            (row["optimal_lambda"], row["achieved_latency"], row["achieved_quality"], row["constraint_satisfied"], row["iterations"])
    for row in a["balanced"]:
        r = LambdaOptimizer(lambda_bounds=(0.1, 20.0)).find_balanced_lambda(analytic, quality_weight=row["quality_weight"])
        assert abs(r.optimal_lambda - row["optimal_lambda"]) <= 1e-9 * row["optimal_lambda"] and r.iterations == row["iterations"]
    assert [list(map(float, t)) for t in LambdaOptimizer().optimize_pareto_front(analytic, 7)] == a["pareto"]
    with pytest.raises(ValueError):
        LambdaOptimizer().optimize_for_latency_constraint(analytic)


def test_grid_search_population_table(oracle_backend, golden):
    from asd_amd.algorithms import GridSearchOptimizer, StagePopulation
    g = golden.npz("lambda_sweep.npz")
    pop = StagePopulation(g["p"], g["C"], ms_per_cost=100.0)
    t = GridSearchOptimizer(lambda_grid=[float(v) for v in g["lam"]]).search_population(pop)
    assert t["costs"] == [float(np.mean(c)) for c in g["cost"]]
    assert t["stage_distributions"][-1] == np.bincount(g["k_star"][-1], minlength=4).tolist()
    assert len(t["latencies"]) == len(g["lam"]) and t["latencies"][0] == t["costs"][0] * 100.0


def test_a14_features_from_device_statistics_equal_the_host_extractor():
    """FeatureExtractor.extract_device (inputs: what asd_verify_accept_stats leaves on the device) == FeatureExtractor.extract
    (the doc's per-token Python, RESEARCH_PROTOCOL.md:366-409) when the per-token lists are whole log-prob vectors."""
    import torch
    from asd_amd.serving.components import FeatureExtractor
    B, T, V = 4, 45, 64
    lps = [[np.log(np.random.default_rng(b * 100 + t).dirichlet(np.ones(V))) for t in range(T)] for b in range(B)]
    n_valid = [45, 33, 7, 0]
    fx = FeatureExtractor()
    host = np.stack([fx.extract("a b c d", "x y z", lps[b][:n_valid[b]], 3) for b in range(B)])
    mx = torch.tensor([[lp.max() for lp in lps[b]] for b in range(B)], dtype=torch.float32)
    en = torch.tensor([[-(np.exp(lp) * lp).sum() for lp in lps[b]] for b in range(B)], dtype=torch.float32)
    dev = FeatureExtractor.extract_device(mx, en, [4] * B, [3] * B, 3, n_valid=torch.tensor(n_valid))
    np.testing.assert_allclose(dev.numpy(), host, atol=2e-6, rtol=0)
    assert dev.shape == (B, 256) and dev[3, 3] == -10.0 and dev[3, 0] == 0.0


def test_hip_decoder_has_no_cpu_form():
    """X3: the HIP decoder stack refuses a CPU model (and the torch modules stay what a CPU model runs)."""
    import pytest
    from asd_amd.serving.synthetic_lm import SyntheticLM, tiny
    lm = SyntheticLM(tiny(), device="cpu")
    assert lm.execution == "torch_modules"
    with pytest.raises(RuntimeError):
        lm.enable_hip_layers()
    assert lm.execution == "torch_modules"
