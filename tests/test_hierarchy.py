"""The three-tier stop-or-escalate loop (asd_amd.serving.hierarchy) on CPU: host control flow with the oracle as
`ops` (tests/oracle_backend.py: OracleOps), tiny f32 SyntheticLMs.

  * single process: some sequences stop at tier 1, others escalate to tier 2 (0 < stop rate < 1); every tier's
    verdict, k*, stop flag and the committed stream are re-derived from the recorded inputs; a tier that sat idle
    for several steps catches its KV cache up correctly (its logits equal a from-scratch forward on the full context);
  * world size 2, 3 and 4 over gloo (BASELINE configs[3]: {7B+32B | 72B}, 7B | 32B | 72B, and 72B vocab-sharded over
    two ranks): the committed stream equals the single-process loop's for the same seeds, and only the small messages
    plus one draft row per stop-with-rejection sequence cross ranks.
The reference's stop-or-continue loop is src/serving/pipeline.py:248-266; the DP rule and the Bayes adjustment are
pinned by tests/golden (test_oracle_golden.py)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

V, B, P, NEW, K = 1000, 6, 5, 24, 4
LAM_MIX = 25.0         # with the predictor below: ~70 % of the blocks stop at tier 1, ~30 % escalate


def _model(noise, seed, dtype=torch.float32, device="cpu", vocab=V):
    from asd_amd.serving.synthetic_lm import SyntheticLM, tiny
    m = SyntheticLM(tiny(vocab=vocab), dtype=dtype, device=device, seed=1, logit_scale=4.0)
    if noise:
        g = torch.Generator().manual_seed(seed)
        with torch.no_grad():
            m.lm_head.weight.add_((torch.randn(m.lm_head.weight.shape, generator=g) * noise).to(dtype).to(device))
    return m


def _predictor():
    from asd_amd.minimal_adaptive_decoder import MinimalQualityPredictor
    torch.manual_seed(0)
    pred = MinimalQualityPredictor().eval()
    with torch.no_grad():
        for p in pred.parameters():
            p.mul_(3.0)                       # spread the scores over (0, 1)
    return pred


def _prompt():
    return torch.randint(0, V, (B, P), generator=torch.Generator().manual_seed(7))


def _cfg(lam=LAM_MIX, min_stage=1):
    from asd_amd.serving import hierarchy as H
    return H.HierarchyConfig(draft_len=K, temperature=0.7, top_p=0.9, lambda_value=lam, seed=3, min_verify_stage=min_stage)


def _single(lam=LAM_MIX, min_stage=1, heads=("logits", "logits"), keep=True):
    from asd_amd.serving import hierarchy as H
    from tests.oracle_backend import OracleOps
    cfg, ops, pred, prompt = _cfg(lam, min_stage), OracleOps(), _predictor(), _prompt()
    d = H.DraftRole(_model(0, 0), cfg, ops, prompt, NEW, pred)
    ts = []
    for s, (noise, seed, hd) in enumerate(zip((0.02, 0.04), (5, 6), heads), start=1):
        m = _model(noise, seed)
        head = H.FusedHead(m, ops) if hd == "fused" else H.LogitsHead(m, ops)
        ts.append(H.VerifyRole(m, s, cfg, ops, prompt, NEW, pred, head=head, keep_inputs=keep))
    return H.generate_hierarchical(d, ts, keep_inputs=keep), cfg, prompt


def test_blocks_stop_or_escalate_and_every_decision_replays():
    from oracle import oracle as O
    from tests.oracle_backend import oracle_predictor_stop
    tr, cfg, prompt = _single()
    assert tr.tier_counts[1] > 0 and tr.tier_counts[2] > 0, tr.tier_counts          # 0 < stop rate < 1
    assert tr.tier_calls[2] < tr.tier_calls[1]                                       # tier 2 only saw the escalated blocks
    assert (tr.seq_len == P + NEW).all()
    pred = _predictor().weights_numpy()
    fresh = {1: _model(0.02, 5), 2: _model(0.04, 6)}
    toks = prompt.to(torch.int32).clone()
    lens = np.full(B, P)
    buf = np.zeros((B, P + NEW), np.int32)
    buf[:, :P] = prompt.numpy()
    costs = np.array(cfg.stage_costs)
    for rec in tr.records:
        dm, final = rec["draft"], rec["final"]
        expect_active = np.ones(B, bool)
        p_prev = dm.p0.numpy()[:, None]
        for s in (1, 2):
            if s not in rec["tiers"]:
                assert not expect_active.any() or s == 2
                continue
            v, drawn = rec["tiers"][s]
            idx = v.idx.numpy()
            assert np.array_equal(idx, np.nonzero(expect_active)[0])                 # the escalation chain
            inp = v.inputs
            n = len(idx)
            # (1) the tier's logits are what a from-scratch forward on committed + drafted tokens gives (KV catch-up)
            for i, b in enumerate(idx):
                ctx = np.concatenate([buf[b, :lens[b]], inp["tok"][i].numpy()])
                fresh[s].reset()
                full = fresh[s](torch.from_numpy(ctx[None, :]).to(torch.int64))[0]
                want = full[lens[b] - 1: lens[b] - 1 + K]
                assert (want - inp["logits"][i]).abs().max().item() < 5e-4
            # (2) verdict on the recorded inputs
            ref = O.verify_accept(inp["logits"].numpy().reshape(n * K, V), O.DT_F32, inp["tok"].numpy(), inp["lp_d"].numpy(),
                                  inp["u"].numpy(), n, K, V, inv_temperature=np.float32(1 / 0.7))
            assert np.array_equal(v.accept.numpy(), ref["accept"])
            assert np.array_equal(v.n_acc.numpy()[idx], ref["n_acc"])
            # (3) stop rule: p_hist carried from the tiers below, this tier's column from its own log-probs
            ph = np.ones((n, 3))
            ph[:, :s] = p_prev[idx, :s]
            score, k_star, hist = oracle_predictor_stop(pred, ref["lp_t"], inp["feat"].numpy(), ph, s, costs, cfg.lambda_value)
            assert np.array_equal(v.k_star.numpy(), k_star)
            assert hist.tobytes() == v.p_hist.numpy().tobytes()
            stop = np.ones(n, bool) if s == 2 else (k_star <= s)
            assert np.array_equal(v.stop.numpy()[idx].astype(bool), stop)
            for i, b in enumerate(idx):                                             # the DP rule itself, CPython form
                ks, _ = O.py_optimal_stopping_rule(list(hist[i]), list(costs), cfg.lambda_value)
                assert ks == k_star[i]
            nxt = np.ones((B, s + 1))
            nxt[idx] = hist[:, :s + 1]
            p_prev = nxt
            expect_active = np.zeros(B, bool)
            expect_active[idx[~stop]] = True
            # the committed verdict is the stopping tier's
            for i, b in enumerate(idx):
                if stop[i]:
                    assert final.tier[b] == s and final.n_acc[b] == ref["n_acc"][i] and final.drawn[b] == drawn[b]
        # (4) the token buffer: accepted prefix + drawn token per sequence, cut at the cap
        for b in range(B):
            new = (list(dm.tok[b, :int(final.n_acc[b])].numpy()) + [int(final.drawn[b])])[: max(0, P + NEW - lens[b])]
            buf[b, lens[b]:lens[b] + len(new)] = new
            lens[b] += len(new)
    assert np.array_equal(buf, tr.tokens.numpy())
    assert sum(tr.tier_counts) == tr.steps * B
    assert tr.rows_shipped <= sum(tr.tier_counts)        # at most one draft row per committed block


def test_lambda_moves_the_stop_distribution_and_stage0_cascade():
    low, _, _ = _single(lam=2.0, keep=False)
    high, _, _ = _single(lam=60.0, keep=False)
    assert low.tier_counts[2] == 0 and low.tier_counts[1] == low.steps * B       # cheap errors: tier 1 is always final
    assert high.tier_counts[1] == 0 and high.tier_counts[2] == high.steps * B    # expensive errors: always escalate
    assert high.fed_tokens[1] > 0 and low.fed_tokens[1] == 0
    casc, _, _ = _single(lam=25.0, min_stage=0, keep=False)                        # the reference's pure cascade
    assert casc.tier_counts[0] > 0 and casc.tier_counts[2] > 0
    assert casc.tier_calls[1] < casc.steps * B                                   # blocks committed at stage 0 were never verified
    assert (casc.seq_len == P + NEW).all()


def test_fused_head_tier_matches_logits_tier():
    """A tier that verifies from hidden states (asd_lm_head_verify; no [n,K,V] logits) commits the same stream."""
    a, _, _ = _single(keep=False)
    b, _, _ = _single(heads=("fused", "fused"), keep=False)
    assert torch.equal(a.tokens, b.tokens) and a.tier_counts == b.tier_counts


# ------------------------------------------------------------------------------------------------ multi-rank
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rank_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from asd_amd.serving import hierarchy as H
        from tests.oracle_backend import OracleOps
        want, cfg, prompt = _single(keep=False)          # the single-process stream, recomputed on every rank
        pl = H.Placement.for_world(world)
        ops, pred = OracleOps(), _predictor()
        draft = H.DraftRole(_model(0, 0), cfg, ops, prompt, NEW, pred) if rank == pl.draft else None
        tiers = {}
        group2 = dist.new_group(pl.ranks_of(2)) if len(pl.ranks_of(2)) > 1 else None
        for s, (noise, seed) in enumerate(((0.02, 5), (0.04, 6)), start=1):
            if rank not in pl.ranks_of(s):
                continue
            m = _model(noise, seed)
            head = None
            if len(pl.ranks_of(s)) > 1:                  # this tier's lm_head is vocab-sharded over its ranks
                head = H.ShardedHead(m, ops, V, group=group2)
                m.lm_head.weight = torch.nn.Parameter(m.lm_head.weight[head.v0:head.v1].clone(), requires_grad=False)
            tiers[s] = H.VerifyRole(m, s, cfg, ops, prompt, NEW, pred, head=head)
        tr = H.run_hierarchical_rank(rank, pl, draft, tiers, B, K, 3, V, torch.float32, P + NEW, "cpu")
        assert torch.equal(tr.tokens, want.tokens), "the multi-rank stream differs from the single-process loop"
        assert torch.equal(tr.seq_len, want.seq_len) and tr.steps == want.steps
        assert tr.tier_counts == want.tier_counts and 0 < tr.tier_counts[1] < sum(tr.tier_counts)
        if rank == pl.draft:
            # bytes that crossed a link from the draft rank: the fixed small messages + ONE row per stop-with-rejection
            small = tr.steps * (B * K * 8 + B * 9) * world + tr.steps * 3 * B * 4 * world
            msgs = tr.messages_sent
            pad = 8 * 4 * sum(msgs.values())          # a message is one buffer, every tensor segment padded to 8 bytes
            rows = tr.bytes_sent.get("rows", 0)
            assert rows <= tr.rows_shipped * (V * 4 + 4) * max(1, len(pl.ranks_of(2))) + 16 * msgs.get("rows", 0)
            assert rows <= sum(tr.tier_counts) * (V * 4 + 4) * max(1, len(pl.ranks_of(2))) + 16 * msgs.get("rows", 0)
            other = sum(v for k, v in tr.bytes_sent.items() if k != "rows")
            assert other <= small + pad, (other, small, pad)
            assert tr.bytes_sent.get("draft", 0) > 0
            # ONE backend call per (message, destination) and step: the draft rank sends `draft` to every verify rank, `rows` to
            # the ranks of a tier that ran, `final` to everyone else
            verify_ranks = sorted({r for t in pl.tiers for r in t} - {pl.draft})
            assert msgs["draft"] == tr.steps * len(verify_ranks)
            assert msgs["final"] == tr.steps * len(sorted(({pl.draft} | {r for t in pl.tiers for r in t}) - {pl.draft}))
            assert msgs.get("rows", 0) <= tr.steps * sum(len([r for r in pl.ranks_of(s_) if r != pl.draft]) for s_ in (1, 2))
        elif tr.messages_sent:
            for name, n in tr.messages_sent.items():  # a verify leader: verdict / drawn to the draft rank, escalate to the next tier's ranks
                dests = {"verdict": 1, "drawn": 1, "escalate": len(pl.ranks_of(2))}.get(name, world)
                assert n <= tr.steps * dests, (name, n)
        ret[rank] = "ok"
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 4])
def test_tiers_on_different_ranks_commit_the_single_process_stream(world):
    ctx = mp.get_context("spawn")
    with ctx.Manager() as mgr:
        ret = mgr.dict()
        port = _free_port()
        procs = [ctx.Process(target=_rank_worker, args=(r, world, port, ret)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(240)
        assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
        assert dict(ret) == {r: "ok" for r in range(world)}


def _sharded_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from asd_amd.serving import hierarchy as H
        from tests.oracle_backend import OracleOps
        solos = [dist.new_group([r]) for r in range(world)]             # collective: every rank creates every group
        Bt = 8
        prompt = torch.randint(0, V, (Bt, P), generator=torch.Generator().manual_seed(11))
        cfg = H.HierarchyConfig(draft_len=K, temperature=0.7, top_p=0.9, stage_costs=(1.0, 10.0), lambda_value=1.0, seed=4)

        def run(group, n, r):
            ops, pred = OracleOps(), _predictor()
            b0, b1 = Bt * r // n, Bt * (r + 1) // n
            d = H.DraftRole(_model(0, 0), cfg, ops, prompt[b0:b1], NEW, pred, batch_total=Bt, batch_offset=b0)
            m = _model(0.03, 9)
            head = H.ShardedHead(m, ops, V, group=group)
            m.lm_head.weight = torch.nn.Parameter(m.lm_head.weight[head.v0:head.v1].clone(), requires_grad=False)
            t = H.ShardedTargetRole(m, cfg, ops, prompt[b0:b1], NEW, pred, head, b0, Bt, group=group)
            return H.run_sharded_target_rank(r, n, d, t, "cpu", max_steps=NEW + 4, group=group), head, t, (b0, b1)

        want, _, t1, _ = run(solos[rank], 1, 0)                         # the whole batch on one rank
        got, head, t, (b0, b1) = run(None, world, rank)                 # replicated drafts; the target's WORK cut over the ranks
        # the rank commits its own sequences: bit for bit the one-rank run's rows
        assert torch.equal(got.tokens, want.tokens[b0:b1]) and torch.equal(got.seq_len, want.seq_len[b0:b1])
        assert (got.seq_len == P + NEW).all() and got.steps == want.steps
        accepted = got.verified_tokens - got.steps * (b1 - b0)
        assert accepted > 0                                             # some drafted tokens were accepted
        # the body ran over THIS rank's rows only: B/N * (K+1) positions per step (round 3: the whole batch on every rank)
        assert t.fed_tokens == got.steps * (b1 - b0) * (K + 1) and t1.fed_tokens == want.steps * Bt * (K + 1)
        assert t.st.B == b1 - b0                                        # ... and so did its KV cache
        # exchange volume per rank: hidden states of its rows + triples + one row piece per sequence and step, never [B,K,V]
        D_ = t.m.shape.hidden
        per_step = (head.bytes_exchanged + t.bytes_exchanged) / got.steps
        # (the draw rows: one all-to-all -- a rank sends its V/N columns of the OTHER ranks' sequences only)
        assert per_step <= ((b1 - b0) * (K + 1) * D_ * 4 + (b1 - b0) * (K * 8 + 8) + Bt * K * 12) * (world - 1) + (Bt - (b1 - b0)) * (V // world + 1) * 4 + 64
        ret[rank] = "ok"
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4, 8])          # 8: the rank count of BASELINE configs[4] (one sequence per rank here)
def test_replicated_drafts_with_a_vocab_sharded_target(world):
    """BASELINE configs[4] in miniature: every rank drafts its slice and runs ITS rows through the target body, the
    target's lm_head is split over all ranks (hidden states all-gathered); the committed stream equals the one-rank run's
    and the per-rank body work is B/N * (K+1) positions per step."""
    ctx = mp.get_context("spawn")
    with ctx.Manager() as mgr:
        ret = mgr.dict()
        port = _free_port()
        procs = [ctx.Process(target=_sharded_worker, args=(r, world, port, ret)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(240)
        assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
        assert dict(ret) == {r: "ok" for r in range(world)}


def test_placements_follow_the_reference_yaml():
    from asd_amd.serving.hierarchy import Placement
    assert Placement.for_world(1).tiers == [[0], [0]]
    assert Placement.for_world(2).tiers == [[0], [1]] and Placement.for_world(2).draft == 0
    assert Placement.for_world(4).tiers == [[1], [2, 3]]
    assert Placement.for_world(8).tiers == [[1], [2, 3, 4, 5]]      # tensor_parallel_size 4 (configs/qwen3_models.yaml:46)


def test_rank_driver_alone_with_loopback_transport_equals_the_single_process_loop():
    """World size 1 with Wire(loopback=True): every message between the roles of the one rank goes through the backend
    (a grouped isend + irecv to itself) instead of the direct hand-over.  The committed stream must not change.  The GPU
    twin of this test runs the same thing over a 1-rank RCCL group (tests/test_gpu_hierarchy.py)."""
    from asd_amd.serving import hierarchy as H
    from tests.oracle_backend import OracleOps
    want, cfg, prompt = _single(keep=False)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    try:
        ops, pred = OracleOps(), _predictor()
        pl = H.Placement.for_world(1)
        d = H.DraftRole(_model(0, 0), cfg, ops, prompt, NEW, pred)
        tiers = {}
        for s_, (noise, seed) in enumerate(zip((0.02, 0.04), (5, 6)), start=1):
            m = _model(noise, seed)
            tiers[s_] = H.VerifyRole(m, s_, cfg, ops, prompt, NEW, pred, head=H.LogitsHead(m, ops))
        got = H.run_hierarchical_rank(0, pl, d, tiers, B, K, 3, V, torch.float32, P + NEW, torch.device("cpu"), loopback=True)
        assert torch.equal(got.tokens, want.tokens) and got.tier_counts == want.tier_counts
        looped = {k: v for k, v in got.bytes_sent.items() if k.endswith("(loopback)")}
        assert set(looped) >= {"draft (loopback)", "verdict (loopback)", "drawn (loopback)", "final (loopback)"} and all(v > 0 for v in looped.values())
    finally:
        dist.destroy_process_group()


def test_a_rank_without_a_role_returns_at_once():
    """Placement.for_world(8) uses ranks 0..5 (7B | 32B | 72B over four ranks): ranks 6 and 7 have no role and must
    neither raise nor wait for a message."""
    from asd_amd.serving import hierarchy as H
    pl = H.Placement.for_world(8)
    assert sorted({pl.draft} | {r for t in pl.tiers for r in t}) == [0, 1, 2, 3, 4, 5]
    tr = H.run_hierarchical_rank(7, pl, None, {}, B, K, 3, V, torch.float32, P + NEW, torch.device("cpu"), max_steps=2)
    assert tr.steps == 0 and tr.verified_tokens == 0


def test_required_hip_layers_fail_loudly_on_cpu():
    """build_rank_roles(hip_layers=True) demands the HIP decoder stack: on a CPU device that is an error, never a silent
    return to the torch modules (hip_layers=None picks the torch modules there and says so in `model.execution`)."""
    from asd_amd.serving import hierarchy as H
    from asd_amd.serving.synthetic_lm import tiny
    from tests.oracle_backend import OracleOps
    cfg = _cfg()
    prompt = _prompt()
    with pytest.raises(RuntimeError):
        H.build_rank_roles(0, H.Placement.for_world(1), [tiny(vocab=V)] * 3, cfg, prompt, 8, _predictor(), ops=OracleOps(), hip_layers=True)
