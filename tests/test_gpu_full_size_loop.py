"""BASELINE configs[2] at FULL size inside the GPU suite: Qwen2.5-7B-shape draft + 32B-shape target on ONE MI355X, batch 32,
draft_len 8, vocab 152064, both models on the HIP decoder stack (asd_decoder_forward), the tier step as ONE launch
(asd_verify_accept_fused_ex on the materialised [32, 8, 152064] bf16 logits).  Two steps of the token-level loop; every
verified block is checked against the oracle ON THE KEPT INPUTS:

  * lp_t of all 256 rows against the f64 log-sum-exp of the stored logits (<= 2e-5), the accept mask bit for bit on every row
    whose acceptance margin exceeds ten times that bound, n_acc wherever the whole prefix is margin-safe;
  * the predictor's score on the kernel's own lp_t (<= 1e-5, BASELINE's bar), p_hist = Bayes(score) bit for bit, k* = the DP
    rule on the kernel's p_hist bit for bit (oracle restatement of src/algorithms/dp_solver.py:12-130);
  * the committed stream: tok[:n_acc] + the drawn token, per sequence.

The reference's loop is src/serving/pipeline.py:165-286 (generate -> predict -> bayesian_adjustment -> optimal_stopping_rule).
Model execution is third party there (vLLM / transformers): the random-weight models only FEED the path real-size logits; what
is pinned is everything from the logits tensor onward.  Seeded weights are generated on the device (no checkpoint, no network)."""
import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu

B, K, P, STEPS = 32, 8, 32, 2


def test_configs2_7b_draft_32b_target_full_size_two_steps_match_the_oracle():
    import torch
    from asd_amd.distributed import HipOps
    from asd_amd.minimal_adaptive_decoder import MinimalQualityPredictor
    from asd_amd.serving import hierarchy as H
    from asd_amd.serving.synthetic_lm import QWEN25_SHAPES
    from tests.oracle_backend import oracle_predictor_stop

    free, _ = torch.cuda.mem_get_info()
    if free < 110e9:
        pytest.skip(f"needs ~95 GB of free HBM for the 7B + 32B shapes, {free / 1e9:.0f} GB free")
    dev = torch.device("cuda", 0)
    shapes = [QWEN25_SHAPES["7b"], QWEN25_SHAPES["32b"]]
    V = shapes[0].vocab
    assert V == 152064 and shapes[1].vocab == V
    torch.manual_seed(0)
    pred = MinimalQualityPredictor().eval()
    with torch.no_grad():
        for p_ in pred.parameters():
            p_.mul_(3.0)
    g = torch.Generator(device=dev).manual_seed(5)
    prompt = torch.randint(0, V, (B, P), generator=g, device=dev)
    cfg = H.HierarchyConfig(draft_len=K, temperature=0.7, top_p=0.9, stage_costs=(1.0, 4.5), lambda_value=2.0, seed=5)
    ops = HipOps()
    new_tokens = (STEPS + 3) * (K + 1)
    draft, tiers = H.build_rank_roles(0, H.Placement(0, [[0]]), shapes, cfg, prompt, new_tokens, pred, ops=ops, heads=("logits",),
                                      logit_scale=0.6, seeds=(1, 2), keep_inputs=True, hip_layers=True)
    assert draft.m.execution == "hip_decoder" and tiers[1].m.execution == "hip_decoder"
    tr = H.generate_hierarchical(draft, [tiers[1]], max_steps=STEPS, keep_inputs=True)
    torch.cuda.synchronize()
    assert tr.steps == STEPS and tr.tier_calls[1] == STEPS * B           # every block was verified by the 32B tier

    weights = pred.weights_numpy()
    costs = np.array(cfg.stage_costs)
    inv_t = np.float32(1 / 0.7)
    lens = np.full(B, P)
    checked = accepted = 0
    for rec in tr.records:
        dm, final = rec["draft"], rec["final"]
        v, drawn = rec["tiers"][1]
        assert np.array_equal(v.idx.cpu().numpy(), np.arange(B))
        inp = v.inputs
        tok, lp_d, u = (inp[k_].cpu().numpy() for k_ in ("tok", "lp_d", "u"))
        assert inp["logits"].shape == (B, K, V) and inp["logits"].dtype == torch.bfloat16
        store = inp["logits"].contiguous().view(torch.int16).cpu().numpy().view(np.uint16).reshape(B * K, V)
        ref = O.verify_accept(store, O.DT_BF16, tok, lp_d, u, B, K, V, inv_temperature=inv_t)
        atol = 2e-5
        np.testing.assert_allclose(inp["lp_t"].cpu().numpy(), ref["lp_t64"], atol=atol, rtol=1e-6)
        safe = ref["margin"] >= 10 * atol
        acc = v.accept.cpu().numpy()
        assert np.array_equal(acc[safe], ref["accept"][safe])
        checked += int(safe.sum())
        n_acc = v.n_acc.cpu().numpy()
        for b in range(B):
            if safe[b].all():
                assert n_acc[b] == ref["n_acc"][b]
        accepted += int(n_acc.sum())
        # the stop rule: stage 0 from the draft's own log-probs, stage 1 from the kernel's lp_t
        _, _, h0 = oracle_predictor_stop(weights, dm.lp_d.cpu().numpy(), draft.feat.cpu().numpy(), np.ones((B, 2)), 0, costs, cfg.lambda_value)
        np.testing.assert_allclose(dm.p0.cpu().numpy(), h0[:, 0], atol=2e-5, rtol=0)
        ph = np.ones((B, 2))
        ph[:, 0] = dm.p0.cpu().numpy()
        score, _, _ = oracle_predictor_stop(weights, inp["lp_t"].cpu().numpy(), inp["feat"].cpu().numpy(), ph, 1, costs, cfg.lambda_value)
        got_score = v.score.cpu().numpy()
        np.testing.assert_allclose(got_score, score, atol=1e-5, rtol=0)
        hist = v.p_hist.cpu().numpy()
        assert hist[:, 0].tobytes() == ph[:, 0].tobytes()
        assert hist[:, 1].tobytes() == O.bayes_adjust(got_score.astype(np.float64), cfg.n_obs).tobytes()
        ks, _ = O.optimal_stopping(hist, costs, cfg.lambda_value)
        assert np.array_equal(v.k_star.cpu().numpy(), ks)
        assert (v.stop.cpu().numpy() == 1).all()                          # the last tier's verdict is final
        # the committed stream
        fin_n, fin_d = final.n_acc.cpu().numpy(), final.drawn.cpu().numpy()
        assert np.array_equal(fin_n, n_acc) and np.array_equal(fin_d, drawn.cpu().numpy())
        assert ((fin_d >= 0) & (fin_d < V)).all()
        toks = tr.tokens.cpu().numpy()
        for b in range(B):
            new = list(tok[b, :fin_n[b]]) + [int(fin_d[b])]
            assert list(toks[b, lens[b]:lens[b] + len(new)]) == new
            lens[b] += len(new)
    assert np.array_equal(tr.seq_len.cpu().numpy(), lens)
    assert checked >= 0.97 * STEPS * B * K, f"only {checked} of {STEPS * B * K} rows were margin-safe"
    assert accepted > 0, "the random 7B / 32B pair (logit_scale 0.6) accepted no drafted token at all"
    # nothing was lost on the way: the hand-off workspaces of this thread report a clean status
    ops.check_status()
    assert all(w.status() == 0 for w in ops._ws.values() if hasattr(w, "status"))
