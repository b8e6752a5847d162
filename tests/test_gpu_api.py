"""The reference-API mirrors and the token-level loop on the real backend (HipBackend): every
number below went through libasd_hip.so on the GPU."""
import os

import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def hip_backend():
    import asd_amd
    asd_amd.set_backend(None)                    # lazy HipBackend
    yield
    asd_amd.set_backend(None)


def test_backend_is_hip():
    import asd_amd
    assert asd_amd.get_backend().name == "hip-gfx950"


def test_decoder_and_predictor_module_on_gpu(golden, tmp_path):
    import torch
    import yaml
    from asd_amd.minimal_adaptive_decoder import MinimalAdaptiveDecoder, MinimalQualityPredictor
    g = golden.npz("predictor.npz")
    pred = MinimalQualityPredictor()
    pred.load_state_dict({"net.0.weight": torch.from_numpy(g["w1"]), "net.0.bias": torch.from_numpy(g["b1"]),
                          "net.3.weight": torch.from_numpy(g["w2"]), "net.3.bias": torch.from_numpy(g["b2"])})
    pred.eval()
    x = torch.from_numpy(g["X"])
    np.testing.assert_allclose(pred(x).numpy()[:, 0], g["scores"], atol=1e-5, rtol=0)            # CPU tensor in
    np.testing.assert_allclose(pred(x.cuda()).cpu().numpy()[:, 0], g["scores"], atol=1e-5, rtol=0)  # device tensor in
    cfg = golden.json("decoder_misc.json")["config"]
    path = tmp_path / "models.yaml"
    path.write_text(yaml.safe_dump(cfg))
    dec = MinimalAdaptiveDecoder(str(path), predictor=pred)
    picks = golden.json("threshold_picks.json")
    for lam, rec in picks.items():
        dec.set_lambda(float(lam))
        assert dec._theta_vector().tolist() == rec["theta"]
    dec.set_lambda(0.1)
    res = dec.decode_batch(["hi", "why is the sky blue?", "how do transformers implement attention? why?"])
    theta = dec._theta_vector()
    for r in res:
        want = next(s for s in range(4) if r.quality_estimate >= theta[s] or s == 3)
        assert r.selected_stage == want


def test_pipeline_on_gpu_backend():
    from asd_amd.serving import AdaptiveSpeculativePipeline, PipelineConfig
    from tests.test_host_logic import FakeStageManager, ScriptedPredictor
    sm = FakeStageManager()
    pipe = AdaptiveSpeculativePipeline(sm, ScriptedPredictor(), object(),
                                       PipelineConfig(lambda_value=30.0, stop_rule="full", risk_adjustment=True))
    res = pipe.batch_process(["easy one", "hard one", "mid one"])
    costs = [1.0, 1.6, 4.2, 8.8]
    for r, word in zip(res, ("easy", "hard", "mid")):
        base = {"easy": 0.97, "mid": 0.6, "hard": 0.05}[word]
        probs = []
        for i in range(4):
            p = 1.0 if i == 3 else O.py_bayesian_adjustment(min(0.99, base + 0.2 * i), 100, 1.0, 1.0)
            probs.append(p)
            P = [1.0] * 4
            P[:i + 1] = probs
            k, _ = O.py_optimal_stopping_rule(P, costs, 30.0)
            if k <= i or i == 3:
                break
        assert r.stopped_at_stage == k and r.stage_probabilities == probs
    prefix = AdaptiveSpeculativePipeline(sm, ScriptedPredictor(), object(), PipelineConfig(stop_rule="prefix"))
    assert prefix.process_request("hard q").stopped_at_stage == 0
    pipe.shutdown()
    prefix.shutdown()


def test_token_logprobs_and_features_on_gpu(golden):
    from asd_amd.training import extract_features_batch, token_logprobs
    g = golden.npz("logprob_idiom.npz")
    np.testing.assert_allclose(token_logprobs(g["scores"], g["tok"]), g["logprob"], rtol=1e-6, atol=1e-5)
    f = golden.npz("features_a7.npz")
    meta = golden.json("features_a7_meta.json")
    mds = [{"logprobs": [float(x) for x in f["logprobs"][i, :int(f["n_valid"][i])]],
            "generation_time": m["generation_time"], "completion_tokens": m["completion_tokens"]}
           for i, m in enumerate(meta)]
    got = extract_features_batch([m["prompt"] for m in meta], [m["output"] for m in meta], mds,
                                 [m["stage_id"] for m in meta])
    assert got.tobytes() == f["features"].tobytes()


@pytest.mark.parametrize("B,K", [(1, 4), (3, 6)])
def test_token_level_loop_accept_masks_match_oracle(B, K):
    """BASELINE configs[0]: random-init 2-layer draft + target (hidden 128, vocab 1k), draft_len 4:
    every step's accept mask / n_acc from the HIP path equals the oracle's on the same logits."""
    import torch
    from asd_amd.minimal_adaptive_decoder import MinimalQualityPredictor
    from asd_amd.serving.speculative import SpeculativeVerifier, speculative_generate
    from asd_amd.serving.synthetic_lm import SyntheticLM, tiny
    dev = torch.device("cuda")
    draft = SyntheticLM(tiny(), dtype=torch.bfloat16, device=dev, seed=1, logit_scale=3.0)
    target = SyntheticLM(tiny(layers=2), dtype=torch.bfloat16, device=dev, seed=2, logit_scale=3.0)
    # a target close to the draft so that some tokens are accepted: share most weights
    target.load_state_dict(draft.state_dict())
    with torch.no_grad():
        target.lm_head.weight.add_(torch.randn_like(target.lm_head.weight) * 0.02)
    torch.manual_seed(0)
    pred = MinimalQualityPredictor().eval()
    ver = SpeculativeVerifier(B, K, 1000, predictor=pred, lambda_value=2.0)
    prompt = torch.randint(0, 1000, (B, 7), device=dev)
    feat = torch.zeros((B, 64), device=dev)
    tr = speculative_generate(draft, target, prompt, 24, ver, seed=5, feat=feat, keep_inputs=True)
    assert tr.tokens.shape == (B, 24) and tr.steps >= 24 // (K + 1)
    accepted = 0
    for inp, mask, stop in zip(tr.step_inputs, tr.accept_masks, tr.stop_flags):
        store = inp["logits"].view(torch.int16).cpu().numpy().view(np.uint16).reshape(B * K, 1000)
        ref = O.verify_accept(store, O.DT_BF16, inp["tok"].cpu().numpy(), inp["lp_d"].cpu().numpy(),
                              inp["u"].cpu().numpy(), B, K, 1000)
        ok = ~(ref["margin"] < 1e-4)
        assert np.array_equal(mask.cpu().numpy()[ok], ref["accept"][ok])
        accepted += int(mask.sum())
        assert stop is not None and stop.shape == (B,)
    assert accepted > 0, "the perturbed target should accept at least some drafted tokens"
    assert tr.verified_tokens >= tr.steps * B


def test_kv_rollback_reproduces_full_context_logits():
    import torch
    from asd_amd.serving.synthetic_lm import SyntheticLM, tiny
    m = SyntheticLM(tiny(), dtype=torch.float32, device="cuda", seed=3)
    ids = torch.randint(0, 1000, (2, 12), device="cuda")
    full = m(ids)
    m.truncate(5)
    assert m.cached_len == 5
    again = m(ids[:, 5:])
    assert (full[:, 5:] - again).abs().max().item() < 1e-4


def test_vocab_sharded_verifier_single_rank_nccl():
    """HipOps through VocabShardedVerifier with a 1-rank RCCL group (the 2-rank exchange is covered on CPU)."""
    import torch
    import torch.distributed as dist
    from asd_amd import distributed as D
    from tests.helpers import make_verify_case, to_device_logits
    import socket
    with socket.socket() as sock:                       # any free port: a fixed one may be taken on a shared box
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        case = make_verify_case(4, 8, 40000, O.DT_BF16, seed=13)
        lg = to_device_logits(case["logits"], case["dtype"]).view(4, 8, 40000)
        v = D.VocabShardedVerifier(40000)
        lp, acc, n_acc, bits = v.verify(lg, torch.from_numpy(case["tok"]).cuda(), torch.from_numpy(case["lp_d"]).cuda(),
                                        torch.from_numpy(case["u"]).cuda())
        torch.cuda.synchronize()
        assert np.array_equal(acc.cpu().numpy(), case["ref"]["accept"])
        assert np.array_equal(n_acc.cpu().numpy(), case["ref"]["n_acc"])
        np.testing.assert_allclose(lp.cpu().numpy(), case["ref"]["lp_t64"], atol=1e-5, rtol=1e-6)
        bv = D.BatchShardedVerifier(4)
        out = bv.verify_local(lg, torch.from_numpy(case["tok"]).cuda(), torch.from_numpy(case["lp_d"]).cuda(),
                              torch.from_numpy(case["u"]).cuda())
        assert np.array_equal(bv.gather_n_acc(out[2]).cpu().numpy(), case["ref"]["n_acc"])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("B,K", [(1, 1), (5, 8), (3, 64)])
def test_commit_step_matches_oracle(B, K):
    """N3: asd_commit_step (token buffer append + length update) is integer work: bit-exact vs the oracle,
    including the clamp at max_len and out-of-range n_acc."""
    import torch
    from asd_amd import kernels as Kn
    rng = np.random.default_rng(B * 100 + K)
    T = 40
    tok = rng.integers(0, 1000, (B, K)).astype(np.int32)
    n_acc = rng.integers(-1, K + 2, B).astype(np.int32)          # -1 and K+1 are clamped
    drawn = rng.integers(0, 1000, B).astype(np.int32)
    seq_len = rng.integers(2, T + 1, B).astype(np.int32)           # some rows are full already
    out = rng.integers(0, 1000, (B, T)).astype(np.int32)
    lens_ref, out_ref, nc_ref = O.commit_step(tok, n_acc, drawn, seq_len, out, max_len=T)
    d = lambda a: torch.from_numpy(a.copy()).cuda()  # noqa: E731
    seq_d, out_d, nc_d = d(seq_len), d(out), torch.zeros(B, dtype=torch.int32, device="cuda")
    Kn.commit_step(d(tok), d(n_acc), d(drawn), seq_d, out_d, nc_d, max_len=T)
    torch.cuda.synchronize()
    assert np.array_equal(seq_d.cpu().numpy(), lens_ref)
    assert np.array_equal(out_d.cpu().numpy(), out_ref)
    assert np.array_equal(nc_d.cpu().numpy(), nc_ref)


def test_ragged_loop_commits_per_sequence_and_matches_oracle():
    """N3: the per-sequence loop.  Accept masks vs the oracle on the recorded inputs; the token buffer is
    exactly prompt + per-step (accepted prefix + drawn token); different sequences advance at different rates."""
    import torch
    from asd_amd.serving import synthetic_lm as SL
    from asd_amd.serving.speculative import SpeculativeVerifier, speculative_generate_ragged
    B, K, V, P, NEW = 4, 4, 1000, 6, 24
    torch.manual_seed(0)
    target = SL.SyntheticLM(SL.tiny(vocab=V, hidden=128, layers=2), device="cuda", seed=1, logit_scale=6.0)
    draft = SL.SyntheticLM(SL.tiny(vocab=V, hidden=128, layers=2), device="cuda", seed=1, logit_scale=5.0)   # a close draft
    prompt = torch.randint(0, V, (B, P), device="cuda")
    ver = SpeculativeVerifier(B, K, V)
    tr = speculative_generate_ragged(draft, target, prompt, NEW, ver, temperature=1.0, seed=3, keep_inputs=True, sync_every=2)
    lens = tr.seq_len.cpu().numpy()
    assert (lens == P + NEW).all()
    toks = tr.tokens.cpu().numpy()
    assert np.array_equal(toks[:, :P], prompt.cpu().numpy())
    cur = np.full(B, P)
    total = 0
    for step, inp in enumerate(tr.step_inputs):
        lg = inp["logits"].view(torch.int16).cpu().numpy().view(np.uint16).reshape(B * K, V)
        ref = O.verify_accept(lg, O.DT_BF16, inp["tok"].cpu().numpy(), inp["lp_d"].cpu().numpy(), inp["u"].cpu().numpy(), B, K, V)
        safe = ref["margin"] >= 1e-4
        assert np.array_equal(tr.accept_masks[step].cpu().numpy()[safe], ref["accept"][safe])
        n_acc, drawn, tk = inp["n_acc"].cpu().numpy(), inp["drawn"].cpu().numpy(), inp["tok"].cpu().numpy()
        nc = tr.commits[step].cpu().numpy()
        for b in range(B):
            new = (list(tk[b, :n_acc[b]]) + [drawn[b]])[: max(0, P + NEW - cur[b])]
            assert nc[b] == len(new)
            assert list(toks[b, cur[b]:cur[b] + len(new)]) == new
            cur[b] += len(new)
        total += int(nc.sum())
    assert total == tr.verified_tokens == B * NEW
    per_seq_steps = [sum(1 for c in tr.commits if c[b].item() > 0) for b in range(B)]
    assert tr.steps < NEW and len(set(per_seq_steps)) >= 1


def test_forward_ragged_from_a_hipgraph_matches_the_eager_pass():
    """Plumbing of the loop (bench.py `loop` record): SyntheticLM.enable_graphs() replays full-batch forward_ragged calls
    from a hipGraph that attends over the whole cache.  Same logits as the eager pass (to the rounding of a longer masked
    softmax), new inputs are honoured on every replay, subset feeds stay eager."""
    import torch
    from asd_amd.serving.synthetic_lm import SyntheticLM, tiny
    B, P, cap = 4, 6, 40
    ids = torch.randint(0, 1000, (B, P), device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
    outs = {}
    for mode in ("eager", "graph"):
        m = SyntheticLM(tiny(), dtype=torch.float32, device="cuda", seed=3)
        m.alloc_ragged(B, cap)
        m.forward_ragged(ids, torch.zeros((B,), dtype=torch.int64, device="cuda"), P)
        if mode == "graph":
            m.enable_graphs(True)
        seq = []
        pos = torch.full((B,), P, dtype=torch.int64, device="cuda")
        tok = ids[:, -1:].clone()
        for step in range(5):
            lg = m.forward_ragged(tok, pos, P + step + 1).clone()
            seq.append(lg)
            tok = lg[:, -1].argmax(-1, keepdim=True)
            pos = pos + 1
        sub = m.forward_ragged(tok[:2], pos[:2], cap, rows=torch.tensor([0, 1], device="cuda"))     # subset: eager either way
        outs[mode] = (torch.stack(seq), sub.clone(), len(m._graphs or {}))
    assert outs["graph"][2] == 1 and outs["eager"][2] == 0
    assert (outs["eager"][0] - outs["graph"][0]).abs().max().item() < 2e-4
    assert (outs["eager"][1] - outs["graph"][1]).abs().max().item() < 2e-4
