"""The three-tier stop-or-escalate loop on the GPU: every launch behind HipOps goes through libasd_hip.so
(asd_draft_sample, asd_verify_accept / asd_lm_head_verify, asd_predictor_stop, asd_residual_sample_ex,
asd_commit_step).  BASELINE configs[3] in miniature on one device: tiny draft / tier-1 / tier-2 models.

Checked per step and tier against the oracle ON THE RECORDED INPUTS: accept mask / n_acc (margin-filtered, bit-exact),
the predictor score (<= 1e-5), p_hist (bit-exact Bayes of the kernel's own score), k* (bit-exact DP on the kernel's
p_hist), the stop flag, the escalation chain, the committed stream; plus: some sequences stop at tier 1 and others
escalate.  Reference anchor of the decision: src/serving/pipeline.py:225-266."""
import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu

V, B, P, NEW, K = 1000, 6, 5, 24, 4


def _build(lam, dtype, heads, min_stage=1):
    import torch
    from asd_amd.distributed import HipOps
    from asd_amd.serving import hierarchy as H
    from tests.test_hierarchy import _model, _predictor, _prompt
    cfg = H.HierarchyConfig(draft_len=K, temperature=0.7, top_p=0.9, lambda_value=lam, seed=3, min_verify_stage=min_stage)
    ops, pred, prompt = HipOps(), _predictor(), _prompt().cuda()
    d = H.DraftRole(_model(0, 0, dtype, "cuda"), cfg, ops, prompt, NEW, pred)
    ts = []
    for s, (noise, seed, hd) in enumerate(zip((0.02, 0.04), (5, 6), heads), start=1):
        m = _model(noise, seed, dtype, "cuda")
        head = H.FusedHead(m, ops) if hd == "fused" else H.LogitsHead(m, ops)
        ts.append(H.VerifyRole(m, s, cfg, ops, prompt, NEW, pred, head=head, keep_inputs=True))
    torch.cuda.synchronize()
    return d, ts, cfg, prompt, pred


def _run_mixed(dtype, heads):
    from asd_amd.serving import hierarchy as H
    for lam in (25.0, 22.0, 28.0, 18.0, 32.0):
        d, ts, cfg, prompt, pred = _build(lam, dtype, heads)
        tr = H.generate_hierarchical(d, ts, keep_inputs=True)
        if tr.tier_counts[1] > 0 and tr.tier_counts[2] > 0:
            return tr, cfg, prompt, pred
    raise AssertionError(f"no lambda gave a mixed stop distribution: {tr.tier_counts}")


def _store(t):
    import torch
    if t.dtype == torch.bfloat16:
        return t.contiguous().view(torch.int16).cpu().numpy().view(np.uint16), O.DT_BF16
    return t.float().contiguous().cpu().numpy(), O.DT_F32


@pytest.mark.parametrize("dtype_name,heads", [("float32", ("logits", "logits")), ("bfloat16", ("logits", "fused"))])
def test_three_tier_loop_on_gpu_matches_oracle(dtype_name, heads):
    import torch
    from tests.oracle_backend import oracle_predictor_stop
    from tests.test_hierarchy import _model
    dtype = getattr(torch, dtype_name)
    tr, cfg, prompt, pred = _run_mixed(dtype, heads)
    print("tier_counts", tr.tier_counts, "tier_calls", tr.tier_calls, "fed", tr.fed_tokens, "steps", tr.steps)
    assert 0 < tr.tier_counts[1] < sum(tr.tier_counts) and tr.tier_counts[2] > 0     # 0 < stop rate at tier 1 < 1
    assert tr.tier_calls[2] < tr.tier_calls[1]
    assert (tr.seq_len == P + NEW).all()
    weights = pred.weights_numpy()
    costs = np.array(cfg.stage_costs)
    inv_t = np.float32(1 / 0.7)
    fresh = {1: _model(0.02, 5, dtype, "cuda"), 2: _model(0.04, 6, dtype, "cuda")} if dtype == torch.float32 else None
    lens = np.full(B, P)
    buf = np.zeros((B, P + NEW), np.int32)
    buf[:, :P] = prompt.cpu().numpy()
    checked_masks = 0
    for rec in tr.records:
        dm, final = rec["draft"], rec["final"]
        expect_active = np.ones(B, bool)
        p_prev = dm.p0.cpu().numpy()[:, None]
        # stage 0: the draft tier's own column of p_hist, from ITS log-probs
        sc0, _, h0 = oracle_predictor_stop(weights, dm.lp_d.cpu().numpy(), tr_feat(rec), np.ones((B, 3)), 0, costs, cfg.lambda_value)
        np.testing.assert_allclose(p_prev[:, 0], h0[:, 0], atol=2e-5, rtol=0)
        for s in (1, 2):
            if s not in rec["tiers"]:
                continue
            v, drawn = rec["tiers"][s]
            idx = v.idx.cpu().numpy()
            assert np.array_equal(idx, np.nonzero(expect_active)[0])
            inp, n = v.inputs, len(idx)
            tok, lp_d, u = (inp[k].cpu().numpy() for k in ("tok", "lp_d", "u"))
            if "logits" in inp:
                store, dt = _store(inp["logits"])
                ref = O.verify_accept(store.reshape(n * K, V), dt, tok, lp_d, u, n, K, V, inv_temperature=inv_t)
                atol = 1e-5
            else:                                  # the tier verified from hidden states: f64 GEMM oracle
                hb = inp["hidden"].reshape(n * K, -1).contiguous().view(torch.int16).cpu().numpy().view(np.uint16)
                wmat = tr_models[s].lm_head.weight
                wb = wmat.contiguous().view(torch.int16).cpu().numpy().view(np.uint16)
                ref = O.lm_head_verify(hb, wb, tok, lp_d, u, n, K, float(np.float32(inv_t * 4.0)))
                ref["lp_t"] = ref["lp_t64"].astype(np.float32)
                atol = 2e-4
            np.testing.assert_allclose(inp["lp_t"].cpu().numpy(), ref["lp_t64"], atol=atol, rtol=1e-6)
            safe = ref["margin"] >= 10 * atol
            assert np.array_equal(v.accept.cpu().numpy()[safe], ref["accept"][safe])
            checked_masks += int(safe.sum())
            if fresh is not None:                  # KV catch-up of a tier that sat idle: from-scratch forward on the context
                for i, b in enumerate(idx):
                    ctx = np.concatenate([buf[b, :lens[b]], tok[i]])
                    fresh[s].reset()
                    full = fresh[s](torch.from_numpy(ctx[None, :]).to(torch.int64).cuda())[0]
                    assert (full[lens[b] - 1: lens[b] - 1 + K] - inp["logits"][i]).abs().max().item() < 2e-3
            # stop rule on the kernel's OWN lp_t: score within 1e-5, p_hist column = Bayes(score) and k* bit-exact
            ph = np.ones((n, 3))
            ph[:, :s] = p_prev[idx, :s]
            score, _, _ = oracle_predictor_stop(weights, inp["lp_t"].cpu().numpy(), inp["feat"].cpu().numpy(), ph, s, costs,
                                                cfg.lambda_value)
            got_score = v.score.cpu().numpy()
            np.testing.assert_allclose(got_score, score, atol=1e-5, rtol=0)
            hist = v.p_hist.cpu().numpy()
            assert hist[:, :s].tobytes() == ph[:, :s].tobytes()                       # carried columns untouched
            assert hist[:, s].tobytes() == O.bayes_adjust(got_score.astype(np.float64), cfg.n_obs).tobytes()
            ks, _ = O.optimal_stopping(hist, costs, cfg.lambda_value)
            assert np.array_equal(v.k_star.cpu().numpy(), ks)
            stop = np.ones(n, bool) if s == 2 else (ks <= s)
            assert np.array_equal(v.stop.cpu().numpy()[idx].astype(bool), stop)
            nxt = np.ones((B, s + 1))
            nxt[idx] = hist[:, :s + 1]
            p_prev = nxt
            expect_active = np.zeros(B, bool)
            expect_active[idx[~stop]] = True
            fin_t, fin_n, fin_d = (x.cpu().numpy() for x in (final.tier, final.n_acc, final.drawn))
            n_acc = v.n_acc.cpu().numpy()
            for i, b in enumerate(idx):
                if stop[i]:
                    assert fin_t[b] == s and fin_n[b] == n_acc[b] and fin_d[b] == drawn.cpu().numpy()[b]
                    assert 0 <= fin_d[b] < V
        tokc = dm.tok.cpu().numpy()
        for b in range(B):
            new = (list(tokc[b, :int(final.n_acc[b])]) + [int(final.drawn[b])])[: max(0, P + NEW - lens[b])]
            buf[b, lens[b]:lens[b] + len(new)] = new
            lens[b] += len(new)
    assert np.array_equal(buf, tr.tokens.cpu().numpy())
    assert checked_masks > 50


def tr_feat(rec):
    """prompt features are the same for every step: recompute from the prompt like DraftRole does."""
    from asd_amd.serving.hierarchy import prompt_features
    from tests.test_hierarchy import _prompt
    return prompt_features(_prompt()).numpy()


tr_models = {}


@pytest.fixture(autouse=True)
def _tier_models():
    """The fused-head check needs tier 2's lm_head matrix: same construction as _build."""
    import torch
    from tests.test_hierarchy import _model
    tr_models[1] = _model(0.02, 5, torch.bfloat16, "cuda")
    tr_models[2] = _model(0.04, 6, torch.bfloat16, "cuda")
    yield
    tr_models.clear()


def test_stage0_cascade_and_lambda_extremes_on_gpu():
    import torch
    from asd_amd.serving import hierarchy as H
    for lam, min_stage, check in ((2.0, 1, lambda t: t.tier_counts[2] == 0 and t.tier_counts[1] > 0),
                                  (60.0, 1, lambda t: t.tier_counts[1] == 0 and t.tier_counts[2] > 0),
                                  (2.0, 0, lambda t: t.tier_counts[0] > 0)):
        d, ts, cfg, prompt, pred = _build(lam, torch.bfloat16, ("logits", "logits"), min_stage)
        tr = H.generate_hierarchical(d, ts)
        assert check(tr), (lam, min_stage, tr.tier_counts)
        assert (tr.seq_len == P + NEW).all()


# ---- the multi-rank drivers with the HIP kernels: two / three ranks sharing cuda:0, messages staged over gloo ----------
def _gpu_rank_worker(rank, world, port, mode, ret):
    import os
    import sys
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch
    import torch.distributed as dist
    from tests.test_hierarchy import ROOT, _model, _predictor, _prompt
    sys.path.insert(0, ROOT)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from asd_amd.distributed import HipOps
        from asd_amd.serving import hierarchy as H
        dev = torch.device("cuda", 0)
        dt = torch.bfloat16
        if mode == "tiers":
            cfg = H.HierarchyConfig(draft_len=K, temperature=0.7, top_p=0.9, lambda_value=25.0, seed=3)
            prompt = _prompt().to(dev)

            def build(pl, r, group_ranks):
                ops, pred = HipOps(), _predictor()
                draft = H.DraftRole(_model(0, 0, dt, dev), cfg, ops, prompt, NEW, pred) if r == pl.draft else None
                tiers = {}
                for s, (noise, seed) in enumerate(((0.02, 5), (0.04, 6)), start=1):
                    if r in pl.ranks_of(s):
                        m = _model(noise, seed, dt, dev)
                        tiers[s] = H.VerifyRole(m, s, cfg, ops, prompt, NEW, pred, head=H.LogitsHead(m, ops))
                return draft, tiers
            d1, t1 = build(H.Placement.for_world(1), 0, None)
            want = H.generate_hierarchical(d1, [t1[1], t1[2]])
            pl = H.Placement.for_world(world)
            d, t = build(pl, rank, None)
            got = H.run_hierarchical_rank(rank, pl, d, t, B, K, 3, V, dt, P + NEW, dev)
            assert torch.equal(got.tokens, want.tokens) and got.tier_counts == want.tier_counts
            assert 0 < got.tier_counts[1] < sum(got.tier_counts)
            if rank == pl.draft:
                # (a message is ONE buffer, each tensor segment padded to 8 bytes: + <= 16 bytes per `rows` message)
                assert got.bytes_sent.get("rows", 0) <= got.rows_shipped * (V * 2 + 4) * len(pl.ranks_of(2)) + 16 * got.messages_sent.get("rows", 0)
        else:                                   # replicated drafts + vocab-sharded target (BASELINE configs[4])
            solos = [dist.new_group([r]) for r in range(world)]
            Bt = 8
            g = torch.Generator().manual_seed(11)
            prompt = torch.randint(0, V, (Bt, P), generator=g).to(dev)
            cfg = H.HierarchyConfig(draft_len=K, temperature=0.7, top_p=0.9, stage_costs=(1.0, 10.0), seed=4)

            def run(group, n, r):
                ops, pred = HipOps(), _predictor()
                b0, b1 = Bt * r // n, Bt * (r + 1) // n
                d = H.DraftRole(_model(0, 0, dt, dev), cfg, ops, prompt[b0:b1].contiguous(), NEW, pred, batch_total=Bt, batch_offset=b0)
                m = _model(0.03, 9, dt, dev)
                head = H.ShardedHead(m, ops, V, group=group)
                m.lm_head.weight = torch.nn.Parameter(m.lm_head.weight[head.v0:head.v1].clone(), requires_grad=False)
                t = H.ShardedTargetRole(m, cfg, ops, prompt[b0:b1].contiguous(), NEW, pred, head, b0, Bt, group=group)
                return H.run_sharded_target_rank(r, n, d, t, dev, max_steps=NEW + 4, group=group), t, (b0, b1)
            want, _, _ = run(solos[rank], 1, 0)
            got, t, (b0, b1) = run(None, world, rank)
            # the shard merge order differs from the one-shard run in the last bits of lp_t only: same decisions, same stream
            assert torch.equal(got.tokens, want.tokens[b0:b1]), "sharded-target stream differs from the one-rank run"
            assert (got.seq_len == P + NEW).all()
            assert t.fed_tokens == got.steps * (b1 - b0) * (K + 1)          # the body ran over this rank's rows only
        torch.cuda.synchronize()
        ret[rank] = "ok"
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,mode", [(2, "tiers"), (3, "tiers"), (2, "sharded")])
def test_multi_rank_drivers_with_hip_kernels_on_one_gpu(world, mode):
    """The multi-rank loops with HipOps: `world` processes share cuda:0, the small messages are staged through the host over
    gloo (distributed.host_staged).  Same protocol, same kernels as an RCCL run on `world` GPUs (which a one-GPU box
    cannot host); the committed stream must equal the single-process GPU loop's."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    with ctx.Manager() as mgr:
        ret = mgr.dict()
        procs = [ctx.Process(target=_gpu_rank_worker, args=(r, world, port, mode, ret)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(300)
        alive = [p for p in procs if p.is_alive()]
        for p in alive:
            p.kill()
        assert not alive and all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
        assert dict(ret) == {r: "ok" for r in range(world)}


def test_rank_driver_over_a_one_rank_rccl_group_with_loopback():
    """run_hierarchical_rank + Wire on the `nccl` backend (RCCL) with DEVICE tensors: a 1-rank group and
    Wire(loopback=True), so every message between the roles of the rank is one grouped ncclSend + ncclRecv of the rank to
    itself instead of the direct hand-over.  (Two ranks need two GPUs: RCCL refuses two ranks on one device.)  The
    committed stream must equal generate_hierarchical's."""
    import socket
    import torch
    import torch.distributed as dist
    from asd_amd.distributed import HipOps
    from asd_amd.serving import hierarchy as H
    from tests.test_hierarchy import _model, _predictor, _prompt
    dt, dev = torch.bfloat16, torch.device("cuda", 0)
    cfg = H.HierarchyConfig(draft_len=K, temperature=0.7, top_p=0.9, lambda_value=25.0, seed=3)

    def build():
        ops, pred, prompt = HipOps(), _predictor(), _prompt().cuda()
        d = H.DraftRole(_model(0, 0, dt, dev), cfg, ops, prompt, NEW, pred)
        tiers = {}
        for s, (noise, seed) in enumerate(zip((0.02, 0.04), (5, 6)), start=1):
            m = _model(noise, seed, dt, dev)
            tiers[s] = H.VerifyRole(m, s, cfg, ops, prompt, NEW, pred, head=H.LogitsHead(m, ops))
        return d, tiers
    d1, t1 = build()
    want = H.generate_hierarchical(d1, [t1[1], t1[2]])
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
    try:
        assert not H.host_staged(None)
        d, t = build()
        got = H.run_hierarchical_rank(0, H.Placement.for_world(1), d, t, B, K, 3, V, dt, P + NEW, dev, loopback=True)
        torch.cuda.synchronize()
        assert torch.equal(got.tokens, want.tokens) and got.tier_counts == want.tier_counts
        looped = {k: v for k, v in got.bytes_sent.items() if k.endswith("(loopback)")}
        assert set(looped) >= {"draft (loopback)", "verdict (loopback)", "drawn (loopback)", "final (loopback)"}
        assert all(v > 0 for v in looped.values())
        # the collectives of the sharded-target loop (configs[4]) on the same RCCL communicator, device tensors: all-gather of the
        # hidden states, the all-to-all of the draw-row pieces, the MIN all-reduce of the loop exit
        from asd_amd.distributed import all_gather_any, all_to_all_rows
        x = torch.arange(6 * 8, device=dev, dtype=torch.float32).view(6, 8).to(dt)
        assert torch.equal(torch.cat(all_gather_any(x), 0), x)
        assert torch.equal(all_to_all_rows(x, [6]), x)
        m_ = torch.tensor([41], device=dev, dtype=torch.int64)
        dist.all_reduce(m_, op=dist.ReduceOp.MIN)
        assert int(m_.item()) == 41
    finally:
        dist.destroy_process_group()


def test_three_tier_loop_at_the_production_vocabulary():
    """The same loop with V = 152064 (Qwen2.5) rows: every kernel of the step at the size the path runs at -- nucleus
    select over 152064 logits, verify from [n, K, 152064] bf16 logits and from hidden states, residual draws."""
    import torch
    from asd_amd.distributed import HipOps
    from asd_amd.serving import hierarchy as H
    from tests.test_hierarchy import _model, _predictor
    VV, NEWV = 152064, 12
    cfg = H.HierarchyConfig(draft_len=K, temperature=0.7, top_p=0.9, lambda_value=25.0, seed=3)
    ops, pred = HipOps(), _predictor()
    prompt = torch.randint(0, VV, (B, P), generator=torch.Generator().manual_seed(7)).cuda()
    dt = torch.bfloat16
    d = H.DraftRole(_model(0, 0, dt, "cuda", VV), cfg, ops, prompt, NEWV, pred)
    m1, m2 = _model(0.02, 5, dt, "cuda", VV), _model(0.04, 6, dt, "cuda", VV)
    ts = [H.VerifyRole(m1, 1, cfg, ops, prompt, NEWV, pred, head=H.LogitsHead(m1, ops), keep_inputs=True),
          H.VerifyRole(m2, 2, cfg, ops, prompt, NEWV, pred, head=H.FusedHead(m2, ops), keep_inputs=True)]
    tr = H.generate_hierarchical(d, ts, keep_inputs=True)
    assert (tr.seq_len == P + NEWV).all() and sum(tr.tier_counts) == tr.steps * B
    costs = np.array(cfg.stage_costs)
    inv_t = np.float32(1 / 0.7)
    checked = 0
    lens = np.full(B, P)
    buf = np.zeros((B, P + NEWV), np.int32)
    buf[:, :P] = prompt.cpu().numpy()
    for rec in tr.records:
        dm, final = rec["draft"], rec["final"]
        assert int(dm.tok.min()) >= 0 and int(dm.tok.max()) < VV and float(dm.lp_d.max()) <= 1e-6
        for s in (1, 2):
            if s not in rec["tiers"]:
                continue
            v, drawn = rec["tiers"][s]
            inp, n = v.inputs, v.idx.numel()
            tok, lp_d, u = (inp[k].cpu().numpy() for k in ("tok", "lp_d", "u"))
            if "logits" in inp:
                ref = O.verify_accept(_store(inp["logits"])[0].reshape(n * K, VV), O.DT_BF16, tok, lp_d, u, n, K, VV,
                                      inv_temperature=inv_t, n_threads=8)
                np.testing.assert_allclose(inp["lp_t"].cpu().numpy(), ref["lp_t64"], atol=1e-5, rtol=1e-6)
                safe = ref["margin"] >= 1e-4
                assert np.array_equal(v.accept.cpu().numpy()[safe], ref["accept"][safe])
                checked += int(safe.sum())
            hist = v.p_hist.cpu().numpy()
            ks, _ = O.optimal_stopping(hist, costs, cfg.lambda_value)
            assert np.array_equal(v.k_star.cpu().numpy(), ks)
            assert 0 <= int(drawn.min()) and int(drawn.max()) < VV
        tokc = dm.tok.cpu().numpy()
        for b in range(B):
            new = (list(tokc[b, :int(final.n_acc[b])]) + [int(final.drawn[b])])[: max(0, P + NEWV - lens[b])]
            buf[b, lens[b]:lens[b] + len(new)] = new
            lens[b] += len(new)
    assert np.array_equal(buf, tr.tokens.cpu().numpy()) and checked > 20
