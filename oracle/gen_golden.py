#!/usr/bin/env python3
"""Generate tests/golden/* from the REFERENCE's own files.  Dev-container only.

Runs the reference source unmodified: each needed file under /root/reference/src is loaded by
path (importlib.util.spec_from_file_location) with empty synthetic parent packages, because
`import src` itself raises TypeError at src/core/interfaces.py:466.  The one
function whose module cannot be imported (extract_features: its module needs the absent
`evaluate` package) is compiled from its own FunctionDef node of the reference file's AST.

Nothing here travels to the GPU box except the fixtures it writes (data: inputs + expected
outputs).  Usage:  PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden.py
"""
from __future__ import annotations

import ast
import importlib.util
import json
import os
import sys
import types

import numpy as np
import torch

REF = os.environ.get("ASD_REFERENCE", "/root/reference")
SRC = os.path.join(REF, "src")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
sys.dont_write_bytecode = True


def _load(name: str, path: str):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def load_reference():
    for pkg in ("src", "src.theory", "src.algorithms"):
        m = types.ModuleType(pkg)
        m.__path__ = []
        sys.modules[pkg] = m
    dp = _load("src.algorithms.dp_solver", os.path.join(SRC, "algorithms", "dp_solver.py"))
    th = _load("src.theory.optimal_stopping", os.path.join(SRC, "theory", "optimal_stopping.py"))
    mad = _load("src.minimal_adaptive_decoder", os.path.join(SRC, "minimal_adaptive_decoder.py"))
    return dp, th, mad


def load_extract_features():
    """Compile ONLY the FunctionDef `extract_features` (generate_training_data.py:148-205)."""
    path = os.path.join(SRC, "training", "generate_training_data.py")
    with open(path, "r") as f:
        tree = ast.parse(f.read(), filename=path)
    fn = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "extract_features")
    mod = ast.Module(body=[fn], type_ignores=[])
    ns = {"np": np, "List": list, "Dict": dict}
    from typing import Dict, List  # noqa: F401  (annotations are evaluated at def time)
    ns.update(Dict=Dict, List=List)
    exec(compile(mod, path, "exec"), ns)
    return ns["extract_features"]


def gen_dp(dp, rng):
    N = 1600
    P = np.full((N, 4), np.nan)
    Cm = np.full((N, 4), np.nan)
    Ls = np.zeros(N, np.int32)
    lam = np.zeros(N)
    risk = np.zeros(N, np.int32)
    alpha = np.ones(N)
    beta = np.ones(N)
    ks = np.zeros(N, np.int32)
    J = np.full((N, 5), np.nan)
    cost_sets = [[1.0, 1.6, 4.2, 8.8], [1.0, 2.0, 4.5, 10.0], [1.0, 4.5, 10.0, 20.0]]
    lam_choices = [0.0, 0.01, 0.1, 0.5, 1.0, 2.0, 5.0, 10.0, 100.0]
    for i in range(N):
        L = int(rng.integers(1, 5))
        mode = i % 8
        if mode == 0:
            p = rng.choice([0.0, 0.5, 1.0], size=L)               # ties / degenerate
        elif mode == 1:
            p = np.round(rng.uniform(0, 1, L), 2)                 # OptimalStoppingTable keys
        else:
            p = rng.uniform(0, 1, L)
        if mode == 2:
            p[-1] = 1.0                                           # pipeline.py:241 last stage
        c = (np.array(cost_sets[i % 3][:L]) if mode != 3 else rng.uniform(0.1, 10, L))
        la = float(lam_choices[i % len(lam_choices)]) if mode != 4 else float(rng.uniform(0, 20))
        ra = int(mode == 5)
        al, be = (float(rng.uniform(0.5, 3)), float(rng.uniform(0.5, 3))) if ra else (1.0, 1.0)
        k, Jr = dp.optimal_stopping_rule([float(x) for x in p], [float(x) for x in c], la,
                                         risk_adjustment=bool(ra), alpha=al, beta=be)
        P[i, :L], Cm[i, :L], Ls[i], lam[i], risk[i], alpha[i], beta[i] = p, c, L, la, ra, al, be
        ks[i] = k
        J[i, :L + 1] = Jr
    # compute_expected_cost on the same cases, at k* and at a random stage
    kk = np.array([rng.integers(0, Ls[i]) for i in range(N)], np.int32)
    ec_star = np.array([dp.compute_expected_cost(list(map(float, P[i, :Ls[i]])),
                                                 list(map(float, Cm[i, :Ls[i]])), float(lam[i]),
                                                 int(ks[i])) for i in range(N)])
    ec_rand = np.array([dp.compute_expected_cost(list(map(float, P[i, :Ls[i]])),
                                                 list(map(float, Cm[i, :Ls[i]])), float(lam[i]),
                                                 int(kk[i])) for i in range(N)])
    np.savez(os.path.join(OUT, "dp_rule.npz"), p=P, C=Cm, L=Ls, lam=lam, risk=risk, alpha=alpha,
             beta=beta, k_star=ks, J=J, k_rand=kk, cost_at_kstar=ec_star, cost_at_krand=ec_rand)


def gen_bayes(dp, rng):
    p = np.concatenate([np.linspace(0, 1, 101), rng.uniform(0, 1, 411)])
    n_obs = rng.choice([1, 10, 100, 101, 1000, 12345, 10**6], size=p.size).astype(np.int64)
    al = rng.choice([0.5, 1.0, 2.0, 7.25], size=p.size)
    be = rng.choice([0.5, 1.0, 2.0, 3.5], size=p.size)
    out = np.array([dp.bayesian_adjustment(float(p[i]), int(n_obs[i]), float(al[i]), float(be[i]))
                    for i in range(p.size)])
    np.savez(os.path.join(OUT, "bayes.npz"), p=p, n_obs=n_obs, alpha=al, beta=be, out=out)


def gen_thresholds(th):
    sets = [([0.7, 0.8, 0.85, 0.9], [1.0, 2.0, 4.5, 10.0]),      # defaults :38-43
            ([0.7, 0.85, 0.9], [1.0, 4.5, 10.0]),                # 3-tier 7B/32B/72B
            ([0.55, 0.75, 0.8, 0.95], [1.0, 1.6, 4.2, 8.8]),
            ([0.6, 0.9], [1.0, 10.0]),
            ([0.8], [1.0])]
    lams = [0.0, 0.1, 0.5, 1.0, 2.0, 5.0, 10.0]
    rows = []
    for q, c in sets:
        for la in lams:
            t = th.OptimalStoppingTheory(th.TheoreticalParameters(
                n_stages=len(q), quality_bounds=list(q), cost_ratios=list(c), lambda_param=la))
            pol = t.derive_optimal_policy()
            rows.append(dict(q=q, c=c, lam=la, theta=[float(pol[s]) for s in range(len(q))]))
    # defaults path (quality_bounds=None) + misc closed forms
    t = th.OptimalStoppingTheory(th.TheoreticalParameters(lambda_param=1.0))
    misc = dict(default_theta=[float(t.derive_optimal_policy()[s]) for s in range(4)],
                regret_bound={str(T): float(t.compute_regret_bound(T)) for T in (10, 100, 1000, 12345)},
                sample_complexity=int(t.sample_complexity()))
    ra = th.RegretAnalyzer(t)
    inst = [(s, d, float(ra.compute_instantaneous_regret(s, d)))
            for s in range(4) for d in (0.0, 0.29, 0.3, 0.49, 0.5, 0.69, 0.7, 1.0)]
    misc["instant_regret"] = inst
    misc["cumulative"] = float(ra.compute_cumulative_regret())
    misc["average"] = float(ra.compute_average_regret())
    misc["tve"] = {k: float(v) for k, v in ra.theoretical_vs_empirical().items()}
    with open(os.path.join(OUT, "thresholds.json"), "w") as f:
        json.dump(dict(rows=rows, misc=misc), f, indent=1)


def gen_predictor(th, mad, rng):
    torch.manual_seed(20251004)
    pred = mad.MinimalQualityPredictor().eval()          # parity is eval mode
    sd = {k: v.detach().numpy().copy() for k, v in pred.state_dict().items()}
    X = rng.standard_normal((256, 64)).astype(np.float32)
    X[:128, 3:] = 0.0                                     # A9 layout: only 3 live columns
    X[:128, :3] = np.abs(X[:128, :3])
    X[250:] *= 20.0                                       # saturating rows
    with torch.no_grad():
        scores = pred(torch.from_numpy(X)).numpy().reshape(-1).copy()
        one_by_one = np.array([pred(torch.from_numpy(X[i]).unsqueeze(0)).item() for i in range(256)])
    picks = {}
    for la in (0.0, 0.01, 0.05, 0.1, 0.5, 1.0):
        t = th.OptimalStoppingTheory(th.TheoreticalParameters(lambda_param=la))
        thr = t.derive_optimal_policy()
        n_models = 4
        sel = []
        for i in range(256):                              # minimal_adaptive_decoder.py:153-164
            q = float(one_by_one[i])
            chosen = None
            for stage_idx in range(n_models):
                threshold = thr.get(stage_idx, 0.0)
                if q >= threshold or stage_idx == n_models - 1:
                    chosen = stage_idx
                    break
            sel.append(chosen)
        picks[str(la)] = dict(theta=[float(thr[s]) for s in range(4)], stage=sel)
    np.savez(os.path.join(OUT, "predictor.npz"), w1=sd["net.0.weight"], b1=sd["net.0.bias"],
             w2=sd["net.3.weight"], b2=sd["net.3.bias"], X=X, scores=scores,
             scores_one_by_one=one_by_one.astype(np.float64))
    with open(os.path.join(OUT, "threshold_picks.json"), "w") as f:
        json.dump(picks, f)

    # A9: MinimalQualityPredictor.extract_features with a stand-in tokenizer (the real one is a
    # fetch-by-name loader, unavailable offline): token ids are DATA supplied by the fixture.
    class _Tok:
        def __init__(self, ids):
            self.ids = ids

        def encode(self, prompt, return_tensors="pt"):
            return torch.tensor([self.ids], dtype=torch.int64)

    a9 = []
    prompts = ["", "hello", "What is the capital of France?", "a b c d e f g " * 30,
               "Explain the proof of the Cauchy-Schwarz inequality step by step " * 12]
    for i, pr in enumerate(prompts):
        T = [0, 1, 7, 300, 700][i]
        ids = [int(x) for x in rng.integers(0, max(2, T // 2 + 1), size=T)]
        if T == 0:
            # length==0 branch: entropy = 0 (:60); torch.unique of an empty tensor is fine
            pass
        feats = pred.extract_features(pr, _Tok(ids)).numpy().astype(np.float32)
        a9.append(dict(prompt=pr, ids=ids, features=[float(x) for x in feats]))
    # A12: _estimate_difficulty / _compute_regret (pure functions of their arguments)
    cfg = {"models": {"stages": [
        {"size_label": "7b", "theoretical_quality": 0.7, "relative_cost": 1.0},
        {"size_label": "14b", "theoretical_quality": 0.8, "relative_cost": 2.0},
        {"size_label": "32b", "theoretical_quality": 0.85, "relative_cost": 4.5},
        {"size_label": "72b", "theoretical_quality": 0.9, "relative_cost": 10.0}]}}
    fake = types.SimpleNamespace(config=cfg)
    a12 = []
    texts = ["hi", "why is the sky blue?", "how do transformers implement attention? why?",
             "Please characterize the asymptotically optimal stopping thresholds " * 4,
             "Explain, how? why? internationalization considerations notwithstanding " * 9]
    for tx in texts:
        d = mad.MinimalAdaptiveDecoder._estimate_difficulty(fake, tx)
        regs = [float(mad.MinimalAdaptiveDecoder._compute_regret(fake, s, d)) for s in range(4)]
        a12.append(dict(prompt=tx, difficulty=float(d), regret=regs))
    with open(os.path.join(OUT, "decoder_misc.json"), "w") as f:
        json.dump(dict(a9=a9, a12=a12, config=cfg), f, indent=1)


def gen_features_a7(rng):
    extract_features = load_extract_features()
    prompts = ["What is 2+2?", "def f(x):\n    return x*2  # explain", "why how when where which what",
               "import numpy as np; a<b", "Translate this sentence into French please", ""]
    outputs = ["4", "It doubles x . It doubles x .", "because " * 17, "ok", "Bonjour le monde " * 5, ""]
    N = 96
    LP = np.zeros((N, 128), np.float64)
    nv = np.zeros(N, np.int32)
    F = np.zeros((N, 64), np.float64)
    meta = []
    for i in range(N):
        n = int(rng.choice([0, 1, 2, 3, 4, 5, 7, 8, 9, 15, 16, 17, 31, 32, 33, 64, 100, 127, 128]))
        lp32 = (-np.abs(rng.standard_normal(n)) * rng.choice([0.1, 1.0, 5.0])).astype(np.float32)
        if i % 11 == 0 and n > 2:
            lp32[1] = lp32[0]                                  # duplicates (tie handling in sort)
        lp = [float(x) for x in lp32]                          # .item() of an f32 tensor (:133)
        pi, oi, st = i % len(prompts), (i * 5) % len(outputs), i % 4
        md = {"logprobs": lp, "generation_time": float(rng.uniform(0.0005, 3.0)),
              "completion_tokens": n}
        F[i] = extract_features(prompts[pi], outputs[oi], md, st)
        LP[i, :n] = lp
        nv[i] = n
        meta.append(dict(prompt=prompts[pi], output=outputs[oi], stage_id=st,
                         generation_time=md["generation_time"], completion_tokens=n))
    np.savez(os.path.join(OUT, "features_a7.npz"), logprobs=LP, n_valid=nv, features=F)
    with open(os.path.join(OUT, "features_a7_meta.json"), "w") as f:
        json.dump(meta, f)


def gen_logprob_idiom(rng):
    """generate_training_data.py:128-136 run with torch exactly as written, on small scores."""
    import torch.nn.functional as F
    R, V = 24, 1000
    scores = (rng.standard_normal((R, V)) * 4).astype(np.float32)
    scores[3, 100:900] = -np.inf                                # top-p style masked vocab
    toks = rng.integers(0, V, size=R).astype(np.int64)
    toks[3] = 5
    toks[4] = int(np.argmax(scores[4]))
    out32 = []
    for i in range(R):
        score = torch.from_numpy(scores[i:i + 1])               # outputs.scores[i]: [1, V]
        probs = F.softmax(score[0], dim=-1)
        token_id = torch.tensor(toks[i])
        out32.append(torch.log(probs[token_id]).item())
    np.savez(os.path.join(OUT, "logprob_idiom.npz"), scores=scores, tok=toks.astype(np.int32),
             logprob=np.array(out32, np.float64))


FULL_V = 152064          # Qwen2.5 vocabulary: the size the path actually runs at
FULL_SEED = 20251004


def full_size_row(seed: int, row: int) -> np.ndarray:
    """The f32 score row of the full-size log-prob goldens: regenerated from (seed, row) on the GPU box by the test
    (tests/test_gpu_verify.py) -- the fixture stores seeds and expected values, not 15 MB of scores."""
    return (np.random.default_rng([seed, row]).standard_normal(FULL_V) * 4.0).astype(np.float32)


def gen_logprob_idiom_full():
    """generate_training_data.py:128-136 run with torch exactly as written, at V = 152064, on the scores the
    reference's generate() call (:110-119: temperature=0.7, top_p=0.9) hands to that loop:

      f32          raw f32 scores (no warpers): the idiom on a plain row
      bf16 / f16   scores that went through 16-bit storage (the reference loads its models in fp16, :79-85; HF
                   up-casts the model's logits to f32 before the warpers), divided by T = 0.7 (TemperatureLogitsWarper)
      *_topp       ... and then masked to the top-p = 0.9 nucleus (TopPLogitsWarper: -inf outside); `keep` lists the
                   surviving token ids so that the test can rebuild the masked row exactly

    For every row: the token (inside the nucleus; one row per masked variant deliberately OUTSIDE -> -inf) and the
    logprob `.item()` of the idiom.  HF's own warper classes are used (third party for the reference as well)."""
    import torch.nn.functional as F
    from transformers.generation.logits_process import TemperatureLogitsWarper, TopPLogitsWarper
    temp, topp = TemperatureLogitsWarper(0.7), TopPLogitsWarper(0.9)
    variants = ["f32", "bf16", "f16", "bf16_topp", "f16_topp"]
    per = 6
    rec = dict(variant=[], row=[], tok=[], logprob=[], keep_off=[0], keep=[])
    rng = np.random.default_rng(FULL_SEED)
    for vi, var in enumerate(variants):
        for j in range(per):
            row = vi * 100 + j
            x = full_size_row(FULL_SEED, row)
            if var.startswith("bf16"):
                x = torch.from_numpy(x).to(torch.bfloat16).float().numpy()        # round to nearest even
            elif var.startswith("f16"):
                x = x.astype(np.float16).astype(np.float32)
            score = torch.from_numpy(x.copy())[None, :]                 # outputs.scores[i]: [1, V]
            keep = np.zeros(0, np.int32)
            if var != "f32":
                score = temp(None, score)
            if var.endswith("_topp"):
                score = topp(None, score)
                keep = torch.isfinite(score[0]).nonzero()[:, 0].numpy().astype(np.int32)
            if var.endswith("_topp") and j == per - 1:
                tok = int(np.setdiff1d(np.arange(64), keep)[0])           # a token the warper masked: log(0)
            elif var.endswith("_topp"):
                tok = int(rng.choice(keep))
            elif j % 2 == 0:
                tok = int(np.argmax(x))
            else:
                tok = int(rng.integers(0, FULL_V))
            probs = F.softmax(score[0], dim=-1)                           # :131
            token_id = torch.tensor(tok)
            lp = torch.log(probs[token_id]).item()                        # :133
            rec["variant"].append(vi)
            rec["row"].append(row)
            rec["tok"].append(tok)
            rec["logprob"].append(lp)
            rec["keep"].append(keep)
            rec["keep_off"].append(rec["keep_off"][-1] + keep.size)
    np.savez(os.path.join(OUT, "logprob_idiom_full.npz"), seed=np.int64(FULL_SEED), vocab=np.int64(FULL_V),
             variants=np.array(variants), variant=np.array(rec["variant"], np.int32), row=np.array(rec["row"], np.int32),
             tok=np.array(rec["tok"], np.int32), logprob=np.array(rec["logprob"], np.float64),
             keep=np.concatenate(rec["keep"]).astype(np.int32), keep_off=np.array(rec["keep_off"], np.int64),
             temperature=np.float32(0.7), top_p=np.float32(0.9))


NUCLEUS_SEED = 20251005


def nucleus_row(seed: int, row: int, V: int, scale: float, storage: str) -> np.ndarray:
    """The score row of the top-p goldens as f32 values (after the storage rounding), regenerated from (seed, row) by the
    tests."""
    x = (np.random.default_rng([seed, row]).standard_normal(V) * float(np.float32(scale))).astype(np.float32)   # (the fixture stores scale as f32)
    if storage == "bf16":
        x = torch.from_numpy(x).to(torch.bfloat16).float().numpy()
    elif storage == "f16":
        x = x.astype(np.float16).astype(np.float32)
    return x


def gen_top_p_nucleus():
    """X1: the proposal distribution the reference samples its training tokens from is HF generate()'s
    (generate_training_data.py:110-119: do_sample, temperature=0.7, top_p=0.9), i.e. TemperatureLogitsWarper followed by
    TopPLogitsWarper of the `transformers` package (requirements.txt: transformers>=4.40,<5; the classes of the installed
    5.15.0 are used, their arithmetic -- ascending sort, f32 softmax + cumsum, remove cumulative <= 1 - top_p -- is
    unchanged since 4.x).  Per row: the nucleus those two classes leave (size, smallest surviving raw score = the
    threshold asd_draft_sample reports, number of scores EQUAL to the threshold that the warper removed -- its tie
    handling follows the sort order, the kernel keeps every tie), the f64 log-sum-exp of the surviving scaled scores, and
    how far top_p is from the two cumulative masses that bracket it (rows closer than 1e-5 are not compared exactly)."""
    from transformers.generation.logits_process import TemperatureLogitsWarper, TopPLogitsWarper
    cases = []
    row = 0
    for V in (1000, 32000, 152064):
        for storage in ("f32", "bf16", "f16"):
            for (T, top_p, scale) in ((0.7, 0.9, 3.0), (0.7, 0.9, 1.0), (1.0, 0.5, 4.0), (1.3, 0.95, 2.0), (0.5, 0.3, 6.0),
                                      (0.7, 0.9, 0.05)):
                cases.append((row, V, storage, T, top_p, scale))
                row += 1
    rec = {k: [] for k in ("row", "V", "storage", "T", "top_p", "scale", "n_keep", "thr", "ties_removed", "lse_keep",
                           "margin")}
    for (row, V, storage, T, top_p, scale) in cases:
        x = nucleus_row(NUCLEUS_SEED, row, V, scale, storage)
        score = torch.from_numpy(x.copy())[None, :]
        score = TopPLogitsWarper(float(top_p))(None, TemperatureLogitsWarper(float(T))(None, score))
        keep = torch.isfinite(score[0]).numpy()
        thr = float(x[keep].min())
        ties_removed = int(((x == np.float32(thr)) & ~keep).sum())
        lse_keep = float(torch.logsumexp(score[0].double()[torch.from_numpy(keep)], dim=0))
        # exact (f64) cumulative masses around the cut, for the comparison margin
        z = np.sort(x.astype(np.float64) / np.float64(np.float32(T)))[::-1]
        pr = np.exp(z - z[0])
        pr /= pr.sum()
        cum = np.cumsum(pr)
        k = int(np.searchsorted(cum, np.float64(np.float32(top_p)), side="left"))
        margin = float(min(abs(cum[min(k, V - 1)] - top_p), abs(cum[k - 1] - top_p) if k > 0 else 1.0))
        for key, val in zip(rec, (row, V, storage, T, top_p, scale, int(keep.sum()), thr, ties_removed, lse_keep, margin)):
            rec[key].append(val)
    np.savez(os.path.join(OUT, "top_p_nucleus.npz"), seed=np.int64(NUCLEUS_SEED), row=np.array(rec["row"], np.int32),
             V=np.array(rec["V"], np.int32), storage=np.array(rec["storage"]), T=np.array(rec["T"], np.float32),
             top_p=np.array(rec["top_p"], np.float32), scale=np.array(rec["scale"], np.float32),
             n_keep=np.array(rec["n_keep"], np.int64), thr=np.array(rec["thr"], np.float32),
             ties_removed=np.array(rec["ties_removed"], np.int64), lse_keep=np.array(rec["lse_keep"], np.float64),
             margin=np.array(rec["margin"], np.float64))


SPEC_SEED = 20251006


def spec_case_inputs(seed: int, case: int, K: int, V: int, scale: float, spread: float):
    """Draft logits [K, V], target logits [K + 1, V] (f32), the drafted tokens (inverse CDF of the draft row against a stored
    uniform) -- regenerated from (seed, case) by the tests."""
    rng = np.random.default_rng([seed, case])
    cand = (rng.standard_normal((K, V)) * float(np.float32(scale))).astype(np.float32)
    new = np.empty((K + 1, V), np.float32)
    new[:K] = (cand.astype(np.float64) + rng.standard_normal((K, V)) * float(np.float32(spread))).astype(np.float32)
    new[K] = (rng.standard_normal(V) * float(np.float32(scale))).astype(np.float32)
    pick = rng.uniform(0, 1, K)
    ids = np.empty(K, np.int64)
    for k in range(K):
        z = cand[k].astype(np.float64)
        q = np.exp(z - z.max())
        q /= q.sum()
        ids[k] = min(int(np.searchsorted(np.cumsum(q), pick[k], side="right")), V - 1)
    return cand, new, ids


def gen_speculative_sampling():
    """A5 + the residual draw against transformers' `_speculative_sampling` (generation/utils.py; algorithm 1 of the
    speculative-decoding paper as HF's assisted generation runs it -- the reference itself has no token-level accept test,
    SURVEY F2).  The function is called unmodified; its two random draws are made deterministic from outside:
    `torch.rand_like` returns the fixture's uniforms (r_i, the acceptance test `r_i <= p_i / q_i`), `torch.multinomial`
    records the distribution it is asked to sample from (p' = norm(max(0, p - q)), or p_{n+1} when every draft was
    accepted).  Stored per case: the uniforms, HF's n_matches, and for three uniforms r the token the inverse CDF (in
    vocabulary order, this build's draw rule) selects from HF's p' -- with the distance of r to the nearer CDF edge, so that
    draws an f32 softmax cannot decide are not compared."""
    import transformers.generation.utils as U
    cases = []
    c = 0
    for V in (1000, 32000):
        for K in (4, 8):
            for scale, spread in ((3.0, 0.5), (4.0, 1.5), (2.0, 0.1)):
                for rep in range(2):
                    cases.append((c, K, V, scale, spread))
                    c += 1
    rec = dict(case=[], K=[], V=[], scale=[], spread=[], u=[], u_off=[0], n_matches=[], r=[], tok=[], margin=[])
    rng = np.random.default_rng(SPEC_SEED)
    real_rand_like, real_multinomial = torch.rand_like, torch.multinomial
    for (case, K, V, scale, spread) in cases:
        cand, new, ids = spec_case_inputs(SPEC_SEED, case, K, V, scale, spread)
        # uniforms away from the decision edge (|log u - log(p_i / q_i)| >= 1e-3), as the parity set of A5 requires
        lq = cand.astype(np.float64) - np.log(np.exp(cand.astype(np.float64) - cand.max(1, keepdims=True)).sum(1, keepdims=True)) - cand.max(1, keepdims=True)
        lp = new[:K].astype(np.float64) - np.log(np.exp(new[:K].astype(np.float64) - new[:K].max(1, keepdims=True)).sum(1, keepdims=True)) - new[:K].max(1, keepdims=True)
        ratio = (lp - lq)[np.arange(K), ids]
        u = rng.uniform(0, 1, K)
        for _ in range(100):
            bad = np.abs(np.log(u) - ratio) < 1e-3
            if not bad.any():
                break
            u[bad] = rng.uniform(0, 1, int(bad.sum()))
        u = u.astype(np.float32)
        got = {}

        def fake_rand_like(t, *a, **k):
            return torch.from_numpy(u.copy()).to(t.dtype).reshape(t.shape)

        def fake_multinomial(p, num_samples=1, **k):
            got["p"] = p.detach().clone()
            return torch.zeros((p.shape[0], num_samples), dtype=torch.long)

        torch.rand_like, torch.multinomial = fake_rand_like, fake_multinomial
        try:
            _, n = U._speculative_sampling(torch.from_numpy(ids)[None, :], torch.from_numpy(cand)[None], K,
                                           torch.from_numpy(new)[None], False)
        finally:
            torch.rand_like, torch.multinomial = real_rand_like, real_multinomial
        n = int(n)
        pp = got["p"][0].double().numpy()
        cum = np.cumsum(pp)
        total = cum[-1]
        rs = rng.uniform(0, 1, 3).astype(np.float32)
        for r in rs:
            target = float(r) * total
            t = int(np.searchsorted(cum, target, side="right"))
            while t < V - 1 and pp[t] <= 0.0:
                t += 1
            t = min(t, V - 1)
            lo = cum[t - 1] if t > 0 else 0.0
            rec["r"].append(r)
            rec["tok"].append(t)
            rec["margin"].append(min(target - lo, cum[t] - target) / total)
        rec["case"].append(case); rec["K"].append(K); rec["V"].append(V); rec["scale"].append(scale)
        rec["spread"].append(spread); rec["u"].append(u); rec["u_off"].append(rec["u_off"][-1] + K)
        rec["n_matches"].append(n)
    np.savez(os.path.join(OUT, "speculative_sampling.npz"), seed=np.int64(SPEC_SEED), case=np.array(rec["case"], np.int32),
             K=np.array(rec["K"], np.int32), V=np.array(rec["V"], np.int32), scale=np.array(rec["scale"], np.float32),
             spread=np.array(rec["spread"], np.float32), u=np.concatenate(rec["u"]).astype(np.float32),
             u_off=np.array(rec["u_off"], np.int64), n_matches=np.array(rec["n_matches"], np.int32),
             r=np.array(rec["r"], np.float32).reshape(-1, 3), tok=np.array(rec["tok"], np.int32).reshape(-1, 3),
             margin=np.array(rec["margin"], np.float64).reshape(-1, 3))


SPEC_FULL_SEED = 20261004


def spec_full_rows(seed: int, case: int, K: int, V: int, scale: float, spread: float, storage: str):
    """RAW draft rows [K, V] and target rows [K + 1, V] of a full-vocabulary case as f32 values AFTER the storage rounding
    (bf16 / f16: what a tier's lm_head really hands over), plus the K uniforms the drafted tokens are drawn with --
    regenerated from (seed, case) by the tests (tests/helpers.py::spec_full_cases)."""
    rng = np.random.default_rng([seed, case])
    cand = (rng.standard_normal((K, V)) * float(np.float32(scale))).astype(np.float32)
    new = np.empty((K + 1, V), np.float32)
    new[:K] = (cand.astype(np.float64) + rng.standard_normal((K, V)) * float(np.float32(spread))).astype(np.float32)
    new[K] = (rng.standard_normal(V) * float(np.float32(scale))).astype(np.float32)
    pick = rng.uniform(0, 1, K).astype(np.float32)

    def rnd(x):
        if storage == "bf16":
            return torch.from_numpy(x).to(torch.bfloat16).float().numpy()
        if storage == "f16":
            return x.astype(np.float16).astype(np.float32)
        return x
    return rnd(cand), rnd(new), pick


def gen_speculative_sampling_full():
    """A5 + the residual draw AT THE SIZE AND SETTINGS THE PATH RUNS AT: V = 152064 (Qwen2.5), rows that
    went through bf16 / f16 storage, the reference's sampling parameters (generate_training_data.py:110-119: temperature
    0.7, top_p 0.9).  As in HF's assisted generation the DRAFT scores are warped by TemperatureLogitsWarper(0.7) +
    TopPLogitsWarper(0.9) (the classes of the installed transformers, called unmodified) before they reach
    `_speculative_sampling`; the TARGET scores by TemperatureLogitsWarper(0.7) only -- the tiers of this build verify
    against the target's full softmax(x / T).  Drafted tokens: inverse CDF (vocabulary order) of HF's warped draft
    distribution against stored uniforms.  `_speculative_sampling` is called unmodified with its uniforms supplied and its
    multinomial input recorded, exactly as in gen_speculative_sampling.  Stored per case (a few hundred bytes; rows are
    regenerated from seeds): the drafted tokens, HF's log q(token), the per-row nucleus threshold in RAW score units (the
    smallest surviving score: what asd_draft_sample reports and asd_residual_sample_ex consumes) with the number of equal
    scores the warper's sort order dropped, the acceptance uniforms,
    HF's n_matches, and for three uniforms the inverse-CDF token of HF's p' with its distance to the nearer CDF edge."""
    import transformers.generation.utils as U
    from transformers.generation.logits_process import TemperatureLogitsWarper, TopPLogitsWarper
    V, T, TOP_P = 152064, 0.7, 0.9
    cases = []
    c = 0
    for storage in ("bf16", "f16"):
        for K in (4, 8):
            for scale, spread in ((3.0, 0.5), (4.0, 1.5), (2.0, 0.1), (6.0, 1.0)):
                cases.append((c, K, scale, spread, storage))
                c += 1
    rec = dict(case=[], K=[], scale=[], spread=[], storage=[], off=[0], u=[], ids=[], lq=[], thr=[], n_keep=[], ties_removed=[], pick_margin=[],
               n_matches=[], r=[], tok=[], margin=[])
    rng = np.random.default_rng(SPEC_FULL_SEED)
    real_rand_like, real_multinomial = torch.rand_like, torch.multinomial
    for (case, K, scale, spread, storage) in cases:
        cand, new, pick = spec_full_rows(SPEC_FULL_SEED, case, K, V, scale, spread, storage)
        cand_w = TopPLogitsWarper(TOP_P)(None, TemperatureLogitsWarper(T)(None, torch.from_numpy(cand.copy())))
        new_w = TemperatureLogitsWarper(T)(None, torch.from_numpy(new.copy()))
        keep = torch.isfinite(cand_w).numpy()
        thr = np.array([cand[k][keep[k]].min() for k in range(K)], np.float32)
        lq_all = torch.log_softmax(cand_w.double(), dim=-1).numpy()
        ids = np.empty(K, np.int64)
        pm = np.empty(K)
        for k in range(K):
            q = np.exp(lq_all[k])
            cum = np.cumsum(q)
            target = float(pick[k]) * cum[-1]
            t = int(np.searchsorted(cum, target, side="right"))
            while t < V - 1 and q[t] <= 0.0:
                t += 1
            ids[k] = min(t, V - 1)
            lo = cum[ids[k] - 1] if ids[k] > 0 else 0.0
            pm[k] = min(target - lo, cum[ids[k]] - target) / cum[-1]
        lq = lq_all[np.arange(K), ids]
        lp = torch.log_softmax(new_w[:K].double(), dim=-1).numpy()[np.arange(K), ids]
        ratio = lp - lq
        u = rng.uniform(0, 1, K)
        for _ in range(100):
            bad = np.abs(np.log(u) - ratio) < 1e-3
            if not bad.any():
                break
            u[bad] = rng.uniform(0, 1, int(bad.sum()))
        u = u.astype(np.float32)
        got = {}

        def fake_rand_like(t, *a, **k):
            return torch.from_numpy(u.copy()).to(t.dtype).reshape(t.shape)

        def fake_multinomial(p, num_samples=1, **k):
            got["p"] = p.detach().clone()
            return torch.zeros((p.shape[0], num_samples), dtype=torch.long)

        torch.rand_like, torch.multinomial = fake_rand_like, fake_multinomial
        try:
            _, n = U._speculative_sampling(torch.from_numpy(ids)[None, :], cand_w[None], K, new_w[None], False)
        finally:
            torch.rand_like, torch.multinomial = real_rand_like, real_multinomial
        n = int(n)
        pp = got["p"][0].double().numpy()
        cum = np.cumsum(pp)
        total = cum[-1]
        for r in rng.uniform(0, 1, 3).astype(np.float32):
            target = float(r) * total
            t = int(np.searchsorted(cum, target, side="right"))
            while t < V - 1 and pp[t] <= 0.0:
                t += 1
            t = min(t, V - 1)
            lo = cum[t - 1] if t > 0 else 0.0
            rec["r"].append(r)
            rec["tok"].append(t)
            rec["margin"].append(min(target - lo, cum[t] - target) / total)
        rec["case"].append(case); rec["K"].append(K); rec["scale"].append(scale); rec["spread"].append(spread)
        rec["storage"].append(storage); rec["off"].append(rec["off"][-1] + K)
        rec["u"].append(u); rec["ids"].append(ids.astype(np.int32)); rec["lq"].append(lq); rec["thr"].append(thr)
        rec["n_keep"].append(keep.sum(1).astype(np.int32)); rec["pick_margin"].append(pm); rec["n_matches"].append(n)
        # scores EQUAL to the threshold that the warper's sort order dropped (the kernels keep every tie): rows with such
        # ties have a slightly different nucleus, so their drafted token / log q are not compared with HF's
        rec["ties_removed"].append(np.array([int(((cand[k] == thr[k]) & ~keep[k]).sum()) for k in range(K)], np.int32))
        print(f"  spec_full case {case}: {storage} K={K} scale={scale} spread={spread} nucleus {keep.sum(1).tolist()} "
              f"ties removed {rec['ties_removed'][-1].tolist()} n_matches {n}")
    np.savez(os.path.join(OUT, "speculative_sampling_full.npz"), seed=np.int64(SPEC_FULL_SEED), V=np.int32(V), T=np.float32(T),
             top_p=np.float32(TOP_P), case=np.array(rec["case"], np.int32), K=np.array(rec["K"], np.int32),
             scale=np.array(rec["scale"], np.float32), spread=np.array(rec["spread"], np.float32),
             storage=np.array(rec["storage"]), off=np.array(rec["off"], np.int64), u=np.concatenate(rec["u"]).astype(np.float32),
             ids=np.concatenate(rec["ids"]), lq=np.concatenate(rec["lq"]).astype(np.float64),
             thr=np.concatenate(rec["thr"]).astype(np.float32), n_keep=np.concatenate(rec["n_keep"]),
             ties_removed=np.concatenate(rec["ties_removed"]),
             pick_margin=np.concatenate(rec["pick_margin"]).astype(np.float64), n_matches=np.array(rec["n_matches"], np.int32),
             r=np.array(rec["r"], np.float32).reshape(-1, 3), tok=np.array(rec["tok"], np.int32).reshape(-1, 3),
             margin=np.array(rec["margin"], np.float64).reshape(-1, 3))


def gen_dynamic_lambda():
    """DynamicCostOptimizer._optimize_lambda_parameter (src/serving/dynamic_cost_optimizer.py:425-487) called unbound
    on a stand-in `self` (the class constructor would start its background thread); the reference source runs
    unmodified.  Inputs cover every branch and its boundaries."""
    from types import SimpleNamespace
    mod = _load("ref_dynamic_cost_optimizer", os.path.join(SRC, "serving", "dynamic_cost_optimizer.py"))
    fn = mod.DynamicCostOptimizer._optimize_lambda_parameter
    rng = np.random.default_rng(777)
    cases = []
    lat_grid = [0.0, 100.0, 139.9, 140.0, 200.0, 260.0, 260.1, 900.0]
    q_grid = [0.0, 0.5, 0.8499, 0.85, 0.9, 0.95, 0.951, 1.0]
    c_grid = [0.0, 1.0, 1.19, 1.2, 1.4, 1.6, 1.61, 3.0]
    for i in range(240):
        if i < 64:
            m = dict(avg_latency=lat_grid[i % 8], avg_quality=q_grid[(i // 8) % 8], avg_cost=c_grid[(i * 3) % 8])
        else:
            m = dict(avg_latency=float(rng.uniform(0, 600)), avg_quality=float(rng.uniform(0.5, 1.0)),
                     avg_cost=float(rng.uniform(0.8, 2.2)))
        if i % 17 == 0:
            m.pop("avg_quality")
        lam = float(rng.choice([0.1, 0.12, 1.0, 2.5, 9.95, 10.0]))
        util = [float(x) for x in rng.uniform(0.6, 1.0, 4)]
        if i % 5 == 0:
            util = [0.95, 0.97, 0.99, 0.92]
        rate = float(rng.uniform(0, 50))
        forecast = [float(x) for x in rng.uniform(0, 60, 2)]
        self_ = SimpleNamespace(lambda_adjustment=lam, target_latency=200, min_quality=0.85,
                                load_predictor=SimpleNamespace(get_load_forecast=lambda hours_ahead=2, f=forecast: dict(enumerate(f))))
        state = SimpleNamespace(gpu_utilization=dict(enumerate(util)), request_rate=rate)
        out = float(fn(self_, m, state))
        cases.append(dict(current_lambda=lam, metrics=m, gpu_utilization=util, request_rate=rate, load_forecast=forecast,
                          new_lambda=out))
    with open(os.path.join(OUT, "dynamic_lambda.json"), "w") as f:
        json.dump(cases, f)


def gen_a4(dp, rng):
    tab = dp.OptimalStoppingTable(lambda_values=[0.1, 1.0, 10.0], num_stages=4)
    grid = [[a, b, c, 1.0] for a in (0.2, 0.5, 0.8) for b in (0.3, 0.6, 0.9) for c in (0.4, 0.7)]
    cost = [1.0, 1.6, 4.2, 8.8]
    tab.precompute(cost, grid)
    queries = [([0.2, 0.3, 0.4, 1.0], 1.0), ([0.204, 0.296, 0.401, 1.0], 0.9), ([0.5, 0.9, 0.7, 1.0], 7.0),
               ([0.33, 0.33, 0.33, 1.0], 1.0), ([0.33, 0.66], 0.2), ([0.8, 0.9, 0.7, 1.0], 0.1)]
    look = [dict(p=p, lam=la, k=int(tab.lookup(p, la)), k_nofallback=int(tab.lookup(p, la, False)))
            for p, la in queries]
    ad = dp.AdaptiveStopping(initial_lambda=0.7, confidence_level=0.1)
    upd = []
    for i in range(60):
        st, q, lat = int(rng.integers(0, 4)), float(rng.uniform(0, 1)), float(rng.uniform(50, 4000))
        ad.update_statistics(st, q, lat)
        upd.append([st, q, lat])
    bounds = [[float(x) for x in ad.get_confidence_bounds(s)] for s in range(4)]
    explore = [bool(ad.should_explore(s)) for s in range(4)]
    fresh = dp.AdaptiveStopping()
    with open(os.path.join(OUT, "a4_table_adaptive.json"), "w") as f:
        json.dump(dict(cost=cost, grid=grid, lambdas=[0.1, 1.0, 10.0], lookups=look, updates=upd,
                       counts=[float(x) for x in ad.stage_counts],
                       rewards=[float(x) for x in ad.stage_rewards], bounds=bounds, explore=explore,
                       fresh_bounds=[str(x) for x in fresh.get_confidence_bounds(0)],
                       fresh_explore=bool(fresh.should_explore(2))), f, indent=1)


def _stage_model(dp, P, C, ms_per_cost):
    """evaluate_function(lam) -> (latency_ms, quality) of a request population described by its predicted
    stage probabilities: every request runs the reference DP rule, latency = mean cost * ms_per_cost,
    quality = mean probability that the stage it stops at is accepted (reference functions only)."""
    def evaluate(lam):
        lat, qual = [], []
        for p in P:
            k, _ = dp.optimal_stopping_rule(list(p), list(C), float(lam))
            ok = 1.0
            for i in range(k + 1):
                ok *= p[i]
            lat.append(sum(C[:k + 1]) * ms_per_cost)
            qual.append(ok)
        return float(np.mean(lat)), float(np.mean(qual))
    return evaluate


def gen_optimizer(dp, rng):
    """N4: src/algorithms/optimizer.py (LambdaOptimizer, GridSearchOptimizer) driven by the DP rule."""
    opt = _load("src.algorithms.optimizer", os.path.join(SRC, "algorithms", "optimizer.py"))
    B, L = 96, 4
    P = np.sort(rng.uniform(0.2, 1.0, (B, L)), axis=1)
    P[:, -1] = 1.0
    C = [1.0, 1.6, 4.2, 8.8]
    lam = np.concatenate([np.logspace(-2, 2, 20), [0.0, 3.3, 250.0]])
    ks = np.zeros((lam.size, B), np.int32)
    cost = np.zeros((lam.size, B))
    ok = np.zeros((lam.size, B))
    tot = np.zeros((lam.size, B))
    for g, lv in enumerate(lam):
        for b in range(B):
            k, _ = dp.optimal_stopping_rule(list(P[b]), C, float(lv))
            pb = 1.0
            for i in range(k + 1):
                pb *= P[b, i]
            ks[g, b], cost[g, b], ok[g, b] = k, sum(C[:k + 1]), pb
            tot[g, b] = dp.compute_expected_cost(list(P[b]), C, float(lv), k)
    np.savez(os.path.join(OUT, "lambda_sweep.npz"), p=P, C=np.array(C), lam=lam, k_star=ks, cost=cost, p_ok=ok,
             expected_cost=tot)
    ev = _stage_model(dp, P, C, 100.0)
    out = dict(ms_per_cost=100.0, latency=[], pareto=[], balanced=[])
    for cons in (150.0, 400.0, 900.0, 50.0):
        r = opt.LambdaOptimizer(latency_constraint=cons).optimize_for_latency_constraint(ev)
        out["latency"].append(dict(constraint=cons, optimal_lambda=r.optimal_lambda, achieved_latency=r.achieved_latency,
                                   achieved_quality=r.achieved_quality, constraint_satisfied=bool(r.constraint_satisfied),
                                   iterations=int(r.iterations)))
    out["pareto"] = [list(map(float, t)) for t in opt.LambdaOptimizer(lambda_bounds=(0.05, 50.0)).optimize_pareto_front(ev, 12)]
    for w in (0.5, 0.9):
        r = opt.LambdaOptimizer().find_balanced_lambda(ev, quality_weight=w)
        out["balanced"].append(dict(quality_weight=w, optimal_lambda=float(r.optimal_lambda),
                                    achieved_latency=r.achieved_latency, achieved_quality=r.achieved_quality,
                                    iterations=int(r.iterations)))
    # an analytic population whose latency FALLS with lambda (the monotonicity the bisection assumes)
    def analytic(lam):
        return 2000.0 / (1.0 + lam) + 40.0, 1.0 / (1.0 + 0.3 * lam)
    out["analytic"] = dict(latency=[], balanced=[])
    for cons, tol, iters in ((150.0, 1e-3, 50), (400.0, 1e-6, 50), (900.0, 1e-3, 5), (30.0, 1e-3, 50)):
        r = opt.LambdaOptimizer(latency_constraint=cons).optimize_for_latency_constraint(analytic, tolerance=tol,
                                                                                         max_iterations=iters)
        out["analytic"]["latency"].append(dict(constraint=cons, tolerance=tol, max_iterations=iters,
                                               optimal_lambda=r.optimal_lambda, achieved_latency=r.achieved_latency,
                                               achieved_quality=r.achieved_quality,
                                               constraint_satisfied=bool(r.constraint_satisfied), iterations=int(r.iterations)))
    for w in (0.3, 0.7):
        r = opt.LambdaOptimizer(lambda_bounds=(0.1, 20.0)).find_balanced_lambda(analytic, quality_weight=w)
        out["analytic"]["balanced"].append(dict(quality_weight=w, optimal_lambda=float(r.optimal_lambda),
                                                achieved_latency=r.achieved_latency, achieved_quality=r.achieved_quality,
                                                iterations=int(r.iterations)))
    out["analytic"]["pareto"] = [list(map(float, t)) for t in opt.LambdaOptimizer().optimize_pareto_front(analytic, 7)]
    with open(os.path.join(OUT, "lambda_optimizer.json"), "w") as f:
        json.dump(out, f, indent=1)


def main():
    os.makedirs(OUT, exist_ok=True)
    dp, th, mad = load_reference()
    rng = np.random.default_rng(1234)
    gen_dp(dp, rng)
    gen_bayes(dp, rng)
    gen_thresholds(th)
    gen_predictor(th, mad, rng)
    gen_features_a7(rng)
    gen_logprob_idiom(rng)
    gen_a4(dp, rng)
    gen_optimizer(dp, np.random.default_rng(4321))
    gen_logprob_idiom_full()
    gen_dynamic_lambda()
    gen_top_p_nucleus()
    gen_speculative_sampling()
    gen_speculative_sampling_full()
    print("wrote", sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()
