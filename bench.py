#!/usr/bin/env python3
"""Benchmark of the draft-verify / accept / stop hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c3|c2|c5]

N > 1 runs one process per GPU.  Under `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` the ranks
come from the launcher (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment); started as a plain process
(`python bench.py --gpus N`, WORLD_SIZE unset) bench.py is its own launcher: before anything touches the GPU it starts the
N ranks as child processes, relays rank 0's line and exits non-zero if a rank fails (_spawn_ranks).

Launch modes (--mode): `graph` (default) captures runs of steps in one hipGraph on one stream;
`eager` is plain stream-ordered launches (within 2 % of graph here: the loop is GPU-bound);
`overlap` forks each step's epilogue to a side stream inside the graph -- measured 29 us/step vs
21 us: cross-queue dependencies cost more than the 5 us epilogue they hide, so it is not the
default.  All modes execute exactly the same kernels on the same buffers and time exactly --steps
steps.

A step = ONE pass of the hot path over one batch of synthetic target logits already resident in
HBM, as ONE launch:  asd_verify_accept_fused_ex = gather + log-sum-exp + acceptance test over [B,K,V] bf16, and -- inside
the same kernel, by the wave that completes each sequence -- log-prob statistics -> 64-d features -> 64x32x1 predictor ->
Bayes -> DP stop rule.  (`--two-launch`: asd_verify_accept_ex followed by asd_predictor_stop, the step of rounds 1-2.)
Logits buffers rotate through > 600 MB so the figure is HBM, not Infinity Cache.
Prints ONE JSON line (rank 0; file descriptor 1 is pointed at stderr for everything else, RCCL's warnings included).
`value` = verified tokens / s = sum_b (n_acc[b] + 1) per second, whole job.  N > 1: one process per GPU, every rank
verifies its own batch of the same shape (batch-parallel replicas, no data-path collective; "weak" scaling); the barriers
and the max-over-ranks of the times run over gloo, RCCL carries the one exchange step of the `sharded_verify` sub-record.
Before the W warm-up steps the same step runs untimed for ~120 ms (a fresh box needs that to leave its idle clocks; the
driver's job is 20 steps), and up to 64 steps are captured per graph (a short job is one replay).

roofline: the ALGORITHMIC bytes (B*K*V*2 + 17*B*K + 4*B, +8*B for the ballot word) of the kernel a TIMED step
launches -- by default k_verify<..., FUSED = true>, verify + accept + the in-kernel epilogue in one launch -- divided by its mean
launch duration; `roofline.plain_kernel` repeats the measurement for the plain streaming kernel (FUSED = false: what
--two-launch runs first), which is NOT what `value` was computed from.  Duration = HIP events recorded on the launch
stream around a back-to-back run of the verify kernel alone over the same rotating buffers, taken
right after the timed region in the same process (it includes the inter-kernel gap, so it is a
slight over-estimate; it agrees with rocprofv3's kernel average within ~1 %, see profiles/).
Event PAIRS around single launches are not used: on this stack a pair adds 5-15 us of its own to
a ~16 us kernel (measured in round 1: 19.8 us per pair vs 16.6 us rocprofv3), and recording them
inside the timed region would slow the very loop `value` is computed from.  cpu_baseline: the C oracle (oracle/,
OpenMP over rows) on the host cores, same workload, bounded sample, rank 0 at N = 1 only.

`loop` (N = 1, after the headline measurement): the token-level three-tier loop of BASELINE configs[2] / [3] on synthetic
7B / 32B / 72B-shape models.  Their passes run on the HIP decoder stack (asd_decoder_forward: asd_linear + csrc/decoder.hip,
DESIGN 4.9) with the projection matrices re-laid tile-major in place; ASD_LOOP_TORCH_MODULES=1 runs the torch modules of rounds
1-3a instead, ASD_LOOP_PACK_WEIGHTS=0 keeps the [N][D] layout.  `loop.model_execution` says which code ran per tier.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (B, K, V, description)
    "c2": (8, 8, 152064, "BASELINE configs[1]: batch 8, draft_len 8, vocab 152064 (Qwen2.5), bf16 logits"),
    "c3": (32, 8, 152064, "BASELINE headline: batch 32, draft_len 8, vocab 152064 (Qwen2.5), bf16 logits"),
    "c5": (128, 8, 152064, "BASELINE configs[4] per-node batch: batch 128, draft_len 8, vocab 152064, bf16 logits"),
}
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
N_STAGES = 3                 # 7B / 32B / 72B tiers
STAGE_COSTS = [1.0, 4.5, 10.0]


def algorithmic_bytes(B, K, V, esz=2):
    return B * K * V * esz + B * K * (4 + 4 + 4) + B * K * (4 + 1) + B * 4 + B * 8


def build_inputs(torch, Kmod, B, K, V, nbuf, device, seed):
    """Synthetic inputs per BASELINE.md §3: logits ~ 4*N(0,1) (bf16); tok = target arg-max w.p. 0.7 else
    uniform; lp_d = lp_t + N(0, 0.5) clipped <= 0; u ~ U(0,1)."""
    g = torch.Generator(device=device).manual_seed(seed)
    ws = Kmod.VerifyWorkspace(B, K, V, torch.bfloat16, device)
    bufs = []
    for _ in range(nbuf):
        lg = torch.empty((B, K, V), dtype=torch.bfloat16, device=device)
        for b in range(B):   # row-block generation keeps the f32 temporary small
            lg[b] = (torch.randn((K, V), generator=g, device=device) * 4.0).to(torch.bfloat16)
        amax = lg.argmax(dim=-1).to(torch.int32)
        rnd = torch.randint(0, V, (B, K), generator=g, device=device, dtype=torch.int32)
        pick = torch.rand((B, K), generator=g, device=device) < 0.7
        tok = torch.where(pick, amax, rnd).contiguous()
        zero = torch.zeros((B, K), device=device)
        half = torch.full((B, K), 0.5, device=device)
        lp_t = Kmod.verify_accept(lg, tok, zero, half, ws).lp_target
        lp_d = torch.clamp(lp_t + torch.randn((B, K), generator=g, device=device) * 0.5, max=0.0).contiguous()
        u = torch.rand((B, K), generator=g, device=device).contiguous()
        out = Kmod.VerifyResult(torch.empty((B, K), dtype=torch.float32, device=device),
                                torch.empty((B, K), dtype=torch.uint8, device=device),
                                torch.empty((B,), dtype=torch.int32, device=device),
                                torch.empty((B,), dtype=torch.int64, device=device))
        bufs.append(dict(logits=lg, tok=tok, lp_d=lp_d, u=u, out=out))
    return ws, bufs


def predictor_weights(np):
    rng = np.random.default_rng(20251004)
    w1 = (rng.standard_normal((32, 64)) / 8.0).astype(np.float32)
    b1 = (rng.standard_normal(32) * 0.05).astype(np.float32)
    w2 = (rng.standard_normal((1, 32)) / 5.0).astype(np.float32)
    b2 = np.zeros(1, np.float32)
    return w1, b1, w2, b2


def cpu_baseline(np, torch, buf, B, K, V, weights, feat, budget_s=12.0):
    """The oracle (oracle/asd_oracle.c, f64 accumulation, OpenMP over rows) on the host cores."""
    from oracle import oracle as O

    O.build()
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = max(1, min(avail, 16))   # the GPU box's CPU share for one GPU is 16 cores
    store = buf["logits"].view(torch.int16).cpu().numpy().view(np.uint16).reshape(B * K, V)
    tok = buf["tok"].cpu().numpy()
    lp_d = buf["lp_d"].cpu().numpy()
    u = buf["u"].cpu().numpy()
    w1, b1, w2, b2 = weights
    Cc = np.asarray(STAGE_COSTS)

    def one_pass():
        r = O.verify_accept(store, O.DT_BF16, tok, lp_d, u, B, K, V, n_threads=cores)
        stats = O.logprob_stats(r["lp_t"], None, K)      # all K target log-probs, as the GPU step does
        x = feat.copy()
        x[:, 5:10] = stats.astype(np.float32)
        score = O.mlp_predict(x, w1, b1, w2[0], b2)
        hist = np.ones((B, N_STAGES))
        hist[:, 0] = O.bayes_adjust(score.astype(np.float64), 100)
        O.optimal_stopping(hist, Cc, 1.0)
        return int(r["n_acc"].sum()) + B

    t0 = time.perf_counter()
    tokens = one_pass()
    first = time.perf_counter() - t0
    passes = max(1, min(5000, int(budget_s / max(first, 1e-3))))
    t0 = time.perf_counter()
    total = 0
    for _ in range(passes):
        total += one_pass()
    dt = time.perf_counter() - t0
    # the reference's own idiom (generate_training_data.py:128-134): per-token torch loop, small sample
    import torch.nn.functional as F
    rows = min(16, B * K)
    xs = buf["logits"].reshape(B * K, V)[:rows].float().cpu()
    tk = buf["tok"].reshape(-1)[:rows].cpu()
    torch.set_num_threads(cores)
    t1 = time.perf_counter()
    for i in range(rows):
        probs = F.softmax(xs[i], dim=-1)
        torch.log(probs[tk[i]]).item()
    idiom_row_s = (time.perf_counter() - t1) / rows
    return dict(value=total / dt, unit="tokens/s", cores=cores, kind="port",
                sample=f"{passes} full passes of the same B={B} K={K} V={V} bf16 step through oracle/asd_oracle.c "
                       f"(f64 LSE, OpenMP over rows, {cores} threads) in {dt:.1f} s",
                ms_per_step=1e3 * dt / passes, tokens_per_step=tokens,
                reference_idiom_ms_per_row=1e3 * idiom_row_s,
                reference_idiom_note="generate_training_data.py:128-134 per-token softmax->index->log->.item() "
                                     f"with torch CPU f32, {rows}-row sample")


def load_traffic(fused=False):
    """HBM bytes per verify launch from the last committed PMC pass (profiles/rNN_traffic.json: the plain kernel;
    rNN_traffic_fused.json: the FUSED instantiation), or None."""
    pdir = os.path.join(ROOT, "profiles")
    try:
        import re
        pat = r"r\d+_traffic_fused\.json" if fused else r"r\d+_traffic\.json"
        cands = sorted(f for f in os.listdir(pdir) if re.fullmatch(pat, f))   # the latest round's record
    except OSError:
        return None, None
    if not cands:
        return None, None
    try:
        with open(os.path.join(pdir, cands[-1])) as f:
            d = json.load(f)
        return d, "profiles/" + cands[-1]
    except Exception:
        return None, None


# ------------------------------------------------------------------------------------------------------------
# The token-level LOOP around the hot path: 7B draft -> 32B -> 72B with the optimal-stopping rule live
# (serving/hierarchy.py; BASELINE configs[2..3]).  Model execution is synthetic random-weight Qwen2.5-shape torch
# modules (third-party in the reference): the numbers below are CONTEXT for the kernel-step rate, not the headline.
def _timed_ops(torch):
    """HipOps whose calls are bracketed by HIP events (hot-path share of a loop step; a pair costs a few us)."""
    from asd_amd.distributed import HipOps

    class TimedOps(HipOps):
        def __init__(self):
            super().__init__()
            self.events = {}

        def _t(self, name, fn, *a, **k):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = fn(*a, **k)
            e1.record()
            self.events.setdefault(name, []).append((e0, e1))
            return out

        def totals_ms(self):
            torch.cuda.synchronize()
            out = {k: (sum(a.elapsed_time(b) for a, b in v), len(v)) for k, v in self.events.items()}
            self.events = {}
            return out

    for name in ("verify_accept", "verify_stop", "lm_head_verify", "lm_head_partial", "accept_from_partials", "predictor_stop",
                 "draft_sample", "residual_sample", "commit_step"):
        def wrap(name=name):
            base = getattr(HipOps, name)

            def f(self, *a, **k):
                return self._t(name, lambda: base(self, *a, **k))
            return f
        setattr(TimedOps, name, wrap())
    return TimedOps()


def hierarchy_loop(torch, dist, device, rank, world, shapes, B, K, prompt_len, warmup, steps, heads=("logits", "fused"),
                   logit_scale=0.6, target_stop_rate=0.66, lam=None, seed=5, group=None):
    """Run `warmup` + `steps` steps of the three-tier stop-or-escalate loop under Placement.for_world(world) and
    return the record (rank 0) -- timed like the headline: barrier + synchronize on both sides, max over ranks.
    group: the DATA-path process group (RCCL: the small point-to-point messages and the vocab-sharded tier's all-gathers move
    device tensors over xGMI); barriers and the reductions of the timings use the default group."""
    from asd_amd.minimal_adaptive_decoder import MinimalQualityPredictor
    from asd_amd.serving import hierarchy as H
    from asd_amd.serving.synthetic_lm import QWEN25_SHAPES, tiny

    table = dict(QWEN25_SHAPES, tiny=tiny())
    shp = [table[n] for n in shapes]
    V = shp[0].vocab
    pl = H.Placement.for_world(world)
    torch.manual_seed(0)
    pred = MinimalQualityPredictor().eval()
    with torch.no_grad():
        for p_ in pred.parameters():
            p_.mul_(3.0)                    # random-init predictor, scores spread over (0, 1); no trained checkpoint offline
    new_tokens = (warmup + steps + 3) * (K + 1)
    g = torch.Generator(device=device).manual_seed(seed)
    prompt = torch.randint(0, V, (B, prompt_len), generator=g, device=device)
    cfg = H.HierarchyConfig(draft_len=K, temperature=0.7, top_p=0.9, lambda_value=lam if lam else 1.0e6, seed=seed)
    ops = _timed_ops(torch)
    t0 = time.perf_counter()
    # model execution: the HIP decoder stack (asd_decoder_forward: 9 launches per layer) unless ASD_LOOP_TORCH_MODULES=1 asks
    # for the torch modules of rounds 1-3 (~50 launches per layer) as the comparison
    hip_layers = False if os.environ.get("ASD_LOOP_TORCH_MODULES", "0") == "1" else None
    data_backend = dist.get_backend(group) if (world > 1 and group is not None) else None
    draft, tiers = H.build_rank_roles(rank, pl, shp, cfg, prompt, new_tokens, pred, ops=ops, heads=heads,
                                      logit_scale=logit_scale, seeds=(1, 2, 3), hip_layers=hip_layers,
                                      pack_weights=os.environ.get("ASD_LOOP_PACK_WEIGHTS", "1") == "1", backend=data_backend)
    torch.cuda.synchronize()
    build_s = time.perf_counter() - t0
    L = len(shp)
    red = torch.device("cpu") if (world > 1 and dist.get_backend() == "gloo") else device   # gloo: one-GPU rehearsal
    graphs_on = os.environ.get("ASD_LOOP_GRAPHS", "0") == "1"
    if graphs_on:
        # plumbing experiment (off by default): the draft's one-token passes and tier 1's full-batch pass replayed from
        # hipGraphs.  Measured in round 3: 15.0 ms instead of 15.9 ms per 7B pass -- the passes are NOT launch-bound, they are
        # bound by the library's skinny GEMMs (M = 32: 14.1 GB of weights in 15 ms = 0.94 TB/s); see DESIGN.md section 6
        for role in ([draft] if draft is not None else []) + [tiers[k_] for k_ in sorted(tiers) if k_ == 1]:
            role.m.enable_graphs(True)

    ctrl = {"c": None}                                   # the running lambda controller (created after the probe step)

    def run(n):
        if world == 1:
            return H.generate_hierarchical(draft, [tiers[s] for s in range(1, L)], max_steps=n, controller=ctrl["c"])
        return H.run_hierarchical_rank(rank, pl, draft, tiers, B, K, L, V, torch.bfloat16, prompt_len + new_tokens, device,
                                       max_steps=n, controller=ctrl["c"], group=group)

    def barrier():
        if world > 1:
            dist.all_reduce(torch.zeros(1, device=red))

    active = draft is not None or bool(tiers)
    calib = None
    if lam is None:
        # one probe step with lambda = 1e6 (every block escalates to the top) records p_hist of all B blocks at tier 1;
        # the lambda controller then picks the lambda whose tier-1 stop share is closest to the target
        share = 0.0
        if active:
            if world == 1:
                tr0 = H.generate_hierarchical(draft, [tiers[s] for s in range(1, L)], max_steps=1, keep_inputs=True)
            else:
                tr0 = H.run_hierarchical_rank(rank, pl, draft, tiers, B, K, L, V, torch.bfloat16, prompt_len + new_tokens,
                                              device, max_steps=1, keep_inputs=True, group=group)
        lam_t = torch.zeros(2, dtype=torch.float64, device=red)
        if 1 in tiers and rank == pl.leader(1):
            v1 = tr0.records[0]["tiers"][1][0]
            lam_v, share = H.calibrate_lambda(ops, v1.p_hist, tiers[1].costs, 1, target_stop_rate)
            lam_t[0], lam_t[1] = lam_v, share
        if world > 1:
            dist.broadcast(lam_t, src=pl.leader(1))
        lam_v, share = float(lam_t[0].item()), float(lam_t[1].item())
        for role in ([draft] if draft is not None else []) + list(tiers.values()):
            role.cfg.lambda_value = lam_v
        cfg.lambda_value = lam_v
        calib = {"target_tier1_stop_rate": target_stop_rate, "probe_share": share,
                 "how": "one probe step at lambda=1e6, asd_lambda_sweep over 256 log-spaced lambdas on the probe's p_hist; then a "
                        "RUNNING controller (hierarchy.StopRateController): after every step the same sweep over the p_hist rows "
                        "tier 1 judged during the last 4 steps re-solves lambda for the target share"}
        if 1 in tiers and rank == pl.leader(1):
            ctrl["c"] = H.StopRateController(ops, tiers[1].cfg, tiers[1].costs, 1, target_stop_rate, window=4, every=1)
            ctrl["c"].observe(tr0.records[0]["tiers"][1][0].p_hist)
    if active and warmup:
        run(warmup)
    ops.totals_ms()
    roles = ([draft] if draft is not None else []) + [tiers[k_] for k_ in sorted(tiers)]
    for role in roles:
        role.events = []                                 # HIP events around every model pass (per-tier torch time)
        role.fwd_calls = 0
        if hasattr(role, "fwd_positions"):
            role.fwd_positions = 0
    for t in tiers.values():
        t.fed_tokens = 0
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tr = run(steps) if active else None
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    hot = ops.totals_ms()
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=red)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    model_ms = {}
    for role in roles:
        name = "tier0_draft" if role is draft else f"tier{role.s}"
        model_ms[name] = {"ms": sum(a.elapsed_time(b_) for a, b_ in role.events), "passes": len(role.events)}
        role.events = None
    if rank != pl.draft:
        return None
    hot_ms = sum(v[0] for v in hot.values())
    # ---- roofline of the LOOP step (north star: "absolute numbers and fraction of HBM roofline"): the bytes a step cannot
    # avoid streaming from HBM -- every model pass reads its tier's weights once (all parameters but the embedding table,
    # of which only the fed rows are touched), the K/V of the sequences it attends over, and the logits it writes and the
    # hot path reads back -- over the measured step time, against the 8 TB/s peak.  Counted for the roles THIS rank hosts.
    def stream_bytes(model):
        total = sum(p_.numel() * p_.element_size() for p_ in model.parameters())
        return total - model.embed.weight.numel() * model.embed.weight.element_size()
    esz = 2
    per_tier = {}
    ctx = prompt_len + (warmup + steps // 2) * (K + 1) * 0.6            # mean context length of the timed region (approx.)
    loop_bytes = 0.0
    for i_, role in enumerate(roles):
        shp_ = role.m.shape
        wbytes = stream_bytes(role.m)
        passes = tr.tier_forwards[0] if role is draft else tr.tier_forwards[role.s]
        positions = tr.tier_forward_positions[0] if role is draft else tr.tier_forward_positions[role.s]
        seqs = positions / (1 if role is draft else (K + 1))             # sequences attended over, summed over passes
        kv = 2.0 * shp_.layers * shp_.kv_heads * shp_.head_dim * ctx * esz * (positions if role is draft else seqs)
        logits = 2.0 * positions * V * esz                               # written by the lm_head GEMM, read by the hot path
        if not role is draft and heads[min(role.s - 1, len(heads) - 1)] == "fused":
            logits = 0.0                                                 # asd_lm_head_verify: the logits never reach HBM
        b_ = passes * wbytes + kv + logits
        loop_bytes += b_
        per_tier["tier0_draft" if role is draft else f"tier{role.s}"] = {
            "weights_GB": wbytes / 1e9, "passes": passes, "positions": positions, "bytes_GB": b_ / 1e9,
            "floor_ms_at_8TBs": 1e3 * b_ / 8e12, "model_ms": model_ms.get("tier0_draft" if role is draft else f"tier{role.s}", {}).get("ms")}
    loop_roof = {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS, "achieved": loop_bytes / elapsed / 1e9,
                 "frac": loop_bytes / elapsed / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_step": loop_bytes / max(1, tr.steps),
                 "floor_ms_per_step_at_8TBs": 1e3 * loop_bytes / max(1, tr.steps) / 8e12, "per_tier": per_tier,
                 "what": "bytes of every model pass (weights once per pass + K/V attended + logits written and re-read) / step time; "
                         "model execution (third party in the reference): see `model_execution` -- the HIP decoder stack of DESIGN 4.9 or torch modules; the hot-path kernels are `hot_path_calls`"}
    rec = {
        "tiers": [s.name for s in shp], "placement": {"draft": pl.draft, "tiers": pl.tiers, "ranks": world},
        "heads": list(heads), "batch": B, "draft_len": K, "prompt_len": prompt_len, "steps": tr.steps, "warmup": warmup,
        "verified_tokens": tr.verified_tokens, "seconds": elapsed, "verified_tokens_per_s": tr.verified_tokens / elapsed,
        "ms_per_step": 1e3 * elapsed / max(1, tr.steps), "tokens_per_sequence_step": tr.verified_tokens / max(1, tr.steps * B),
        "lambda": cfg.lambda_value, "lambda_calibration": calib, "stage_costs": list(cfg.stage_costs),
        "tier_counts": tr.tier_counts, "stop_rate": tr.stop_rate, "tier_calls": tr.tier_calls,
        "fed_tokens": tr.fed_tokens, "draft_rows_shipped": tr.rows_shipped, "bytes_sent_from_draft_rank": tr.bytes_sent,
        "bytes_sent": tr.bytes_sent, "messages_sent": getattr(tr, "messages_sent", {}),
        "rccl_ranks": (dist.get_world_size(group) if world > 1 else 1), "backend": (dist.get_backend(group) if world > 1 else None),
        "hot_path_ms_per_step_on_draft_rank": hot_ms / max(1, tr.steps),
        "hot_path_calls": {k: {"ms": v[0], "calls": v[1]} for k, v in hot.items()},
        "hot_path_share": hot_ms / (1e3 * elapsed), "build_s": build_s,
        "roofline": loop_roof, "model_ms": model_ms, "model_passes_from_hipgraphs": graphs_on,
        "lambda_history": tr.lambda_history, "tier_forwards": tr.tier_forwards,
        "model_execution": {("tier0_draft" if r_ is draft else f"tier{r_.s}"): r_.m.execution
                            for r_ in ([draft] if draft is not None else []) + [tiers[k_] for k_ in sorted(tiers)]},
        "models": "synthetic random-weight Qwen2.5 shapes (third-party in the reference), logit_scale "
                  f"{logit_scale} so that unrelated random models still accept tokens; KV per sequence (ragged)",
    }
    return rec


def sharded_target_loop(torch, dist, device, rank, world, shapes, B_local, K, prompt_len, warmup, steps, logit_scale=0.6, seed=5,
                        group=None):
    """BASELINE configs[4]: every rank drafts its own B_local sequences (replicated draft tier) and runs ITS rows through its
    replica of the target body (288 GB hold the 143 GB of a 72B body next to the draft: the WORK is sharded along the batch, the
    per-rank model pass does not grow with N); the target's lm_head is vocabulary-sharded over ALL ranks
    (hierarchy.ShardedTargetRole: all-gather of the final hidden states [B/N, K+1, D], asd_lm_head_partial, [B,K,3] all-gather).
    Weak scaling: the batch grows with the ranks.  group: the data-path group (RCCL); control reductions use the default group."""
    from asd_amd.minimal_adaptive_decoder import MinimalQualityPredictor
    from asd_amd.serving import hierarchy as H
    from asd_amd.serving.synthetic_lm import QWEN25_SHAPES, SyntheticLM, tiny

    table = dict(QWEN25_SHAPES, tiny=tiny())
    d_shape, t_shape = table[shapes[0]], table[shapes[-1]]
    V = t_shape.vocab
    Bt = B_local * world
    b0, b1 = B_local * rank, B_local * (rank + 1)
    torch.manual_seed(0)
    pred = MinimalQualityPredictor().eval()
    new_tokens = (warmup + steps + 3) * (K + 1)
    g = torch.Generator(device=device).manual_seed(seed)
    prompt = torch.randint(0, V, (Bt, prompt_len), generator=g, device=device)
    cfg = H.HierarchyConfig(draft_len=K, temperature=0.7, top_p=0.9, stage_costs=(1.0, 10.0), lambda_value=1.0, seed=seed)
    ops = _timed_ops(torch)
    t0 = time.perf_counter()
    dm = SyntheticLM(d_shape, dtype=torch.bfloat16, device=device, seed=1, logit_scale=logit_scale)
    if os.environ.get("ASD_LOOP_TORCH_MODULES", "0") != "1" and d_shape.head_dim == 128:
        dm.enable_hip_layers()
    draft = H.DraftRole(dm, cfg, ops, prompt[b0:b1].contiguous(), new_tokens, pred, batch_total=Bt, batch_offset=b0)
    tm = SyntheticLM(t_shape, dtype=torch.bfloat16, device=device, seed=3, logit_scale=logit_scale)
    if os.environ.get("ASD_LOOP_TORCH_MODULES", "0") != "1" and t_shape.head_dim == 128:
        tm.enable_hip_layers()
    head = H.ShardedHead(tm, ops, V, group=group)
    tm.lm_head.weight = torch.nn.Parameter(tm.lm_head.weight[head.v0:head.v1].clone(), requires_grad=False)
    torch.cuda.empty_cache()
    target = H.ShardedTargetRole(tm, cfg, ops, prompt[b0:b1].contiguous(), new_tokens, pred, head, b0, Bt, group=group)
    torch.cuda.synchronize()
    build_s = time.perf_counter() - t0
    red = torch.device("cpu") if dist.get_backend() == "gloo" else device

    def barrier():
        dist.all_reduce(torch.zeros(1, device=red))

    if warmup:                                       # also builds the data group's communicator before the timed steps
        H.run_sharded_target_rank(rank, world, draft, target, device, max_steps=warmup, group=group)
    ops.totals_ms()
    head.bytes_exchanged = 0
    target.bytes_exchanged = 0
    target.fed_tokens = 0
    target.fwd_calls = draft.fwd_calls = draft.fwd_positions = 0
    draft.events, target.events = [], []
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tr = H.run_sharded_target_rank(rank, world, draft, target, device, max_steps=steps, group=group)
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    hot = ops.totals_ms()
    t = torch.tensor([elapsed, 0.0], dtype=torch.float64, device=red)
    dist.all_reduce(t[:1], op=dist.ReduceOp.MAX)
    elapsed = float(t[0].item())
    tk = torch.tensor([float(tr.verified_tokens)], dtype=torch.float64, device=red)
    dist.all_reduce(tk, op=dist.ReduceOp.SUM)        # every rank commits its own sequences: the job's tokens are the sum
    verified = int(tk.item())
    model_ms = {"tier0_draft": {"ms": sum(a.elapsed_time(b_) for a, b_ in draft.events), "passes": len(draft.events)},
                "tier1_target": {"ms": sum(a.elapsed_time(b_) for a, b_ in target.events), "passes": len(target.events)}}
    draft.events = target.events = None
    if rank != 0:
        return None
    hot_ms = sum(v[0] for v in hot.values())

    def stream_bytes(model):
        total = sum(p_.numel() * p_.element_size() for p_ in model.parameters())
        return total - model.embed.weight.numel() * model.embed.weight.element_size()
    # per-RANK roofline of the step (weak scaling: the same on every rank): every model pass streams its weights once; the
    # target's lm_head shard is streamed once for the gathered [B, K] rows
    ctx = prompt_len + (warmup + steps // 2) * (K + 1) * 0.6
    per = {}
    rank_bytes = 0.0
    for name, role, passes, positions, seqs in (("tier0_draft", draft, tr.tier_forwards[0], tr.tier_forward_positions[0], tr.tier_forward_positions[0]),
                                                ("tier1_target", target, tr.tier_forwards[1], tr.tier_forward_positions[1],
                                                 tr.tier_forward_positions[1] / (K + 1))):
        shp_ = role.m.shape
        kv = 2.0 * shp_.layers * shp_.kv_heads * shp_.head_dim * ctx * 2 * seqs
        b_ = passes * stream_bytes(role.m) + kv + (2.0 * positions * V * 2 if role is draft else 0.0)
        rank_bytes += b_
        per[name] = {"weights_GB": stream_bytes(role.m) / 1e9, "passes": passes, "positions": positions, "bytes_GB": b_ / 1e9,
                     "floor_ms_at_8TBs": 1e3 * b_ / 8e12, "model_ms": model_ms[name]["ms"]}
    roof = {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS, "achieved": rank_bytes / elapsed / 1e9,
            "frac": rank_bytes / elapsed / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_step": rank_bytes / max(1, tr.steps),
            "floor_ms_per_step_at_8TBs": 1e3 * rank_bytes / max(1, tr.steps) / 8e12, "per_tier": per,
            "what": "PER RANK (rank 0): bytes of every model pass of this rank (weights once per pass + K/V attended + the draft's "
                    "logits written and re-read; the target's logits never reach HBM) / step time"}
    return {
        "tiers": [d_shape.name, t_shape.name],
        "placement": f"replicated {d_shape.name} drafts on every rank; {t_shape.name} target: body replicated, its WORK sharded along "
                     f"the batch ({B_local} sequences per rank), lm_head vocab-sharded over {world} rank(s)",
        "rccl_ranks": dist.get_world_size(group), "backend": dist.get_backend(group),
        "batch_total": Bt, "batch_per_rank": B_local,
        "draft_len": K, "prompt_len": prompt_len, "steps": tr.steps, "warmup": warmup, "verified_tokens": verified,
        "seconds": elapsed, "verified_tokens_per_s": verified / elapsed, "ms_per_step": 1e3 * elapsed / max(1, tr.steps),
        "tokens_per_sequence_step": verified / max(1, tr.steps * Bt),
        "fed_tokens_per_rank_per_step": tr.fed_tokens[0] / max(1, tr.steps),
        "bytes_sent": {"hidden_states+drafts": target.bytes_exchanged, "triples+row_pieces": head.bytes_exchanged},
        "bytes_exchanged_per_step_rank0": (head.bytes_exchanged + target.bytes_exchanged) / max(1, tr.steps),
        "hot_path_ms_per_step": hot_ms / max(1, tr.steps), "hot_path_share": hot_ms / (1e3 * elapsed),
        "hot_path_calls": {k: {"ms": v[0], "calls": v[1]} for k, v in hot.items()}, "build_s": build_s,
        "roofline": roof, "model_ms": model_ms,
        "model_execution": {"tier0_draft": draft.m.execution, "tier1_target": target.m.execution},
        "models": f"synthetic random-weight Qwen2.5 shapes, logit_scale {logit_scale}",
    }


LM_HEADS = {"7b": 3584, "14b": 5120, "32b": 5120, "72b": 8192}   # Qwen2.5 hidden sizes (configs/models.yaml)
MFMA_PEAK_TFLOPS = 2500.0    # MI355X_MICROARCH.md: dense bf16


def main_lm_head(args):
    """`--lm-head SIZE`: the step starts one stage earlier -- from the target's final hidden
    states [B, K, D] and its lm_head matrix [V, D]: asd_lm_head_verify (bf16 MFMA GEMM + log-sum-exp + accept,
    logits never in HBM) + asd_predictor_stop.  Same contract as the default bench; roofline bound = mfma."""
    import numpy as np
    import torch
    import torch.distributed as dist

    from asd_amd import kernels as Kmod

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if os.environ.get("ASD_BENCH_ONE_DEVICE") == "1" else int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} does not match the launcher's WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    red_dev = device if args.dist_backend == "nccl" else torch.device("cpu")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(args.dist_backend, **({"device_id": device} if args.dist_backend == "nccl" else {}))

    def barrier():
        if world > 1:
            dist.all_reduce(torch.zeros(1, device=red_dev))

    B, K, V, desc = WORKLOADS[args.workload]
    D = LM_HEADS[args.lm_head]
    M = B * K
    g = torch.Generator(device=device).manual_seed(4321 + rank)
    w = (torch.randn((V, D), device=device, generator=g) * (3.0 / D ** 0.5)).to(torch.bfloat16)
    nbuf = 4
    bufs = []
    for _ in range(nbuf):
        h = torch.randn((M, D), device=device, generator=g).to(torch.bfloat16)
        tok = torch.randint(0, V, (B, K), device=device, generator=g, dtype=torch.int32)
        tok[:, ::2] = (h[::2].float() @ w.float().T).argmax(-1).reshape(B, -1)[:, : tok[:, ::2].shape[1]].to(torch.int32)
        bufs.append(dict(h=h, tok=tok, lp_d=-torch.rand((B, K), device=device, generator=g) * 2,
                         u=torch.rand((B, K), device=device, generator=g), out=None))
    ver = Kmod.LmHeadVerifier(w, B, K, packed=True)      # tile-major copy of the matrix (asd_lm_head_pack_weights)
    packed = Kmod.pack_mlp_weights(*predictor_weights(np), device=device)
    feat = torch.from_numpy((np.random.default_rng(7).standard_normal((B, 64)) * 0.3).astype(np.float32)).to(device)
    Cc = torch.tensor(STAGE_COSTS, dtype=torch.float64, device=device)
    p_hist = torch.ones((B, N_STAGES), dtype=torch.float64, device=device)

    def step(i):
        b = bufs[i % nbuf]
        b["out"] = ver(b["h"], b["tok"], b["lp_d"], b["u"], out=b["out"])
        if not args.verify_only:
            Kmod.predictor_stop(feat, packed, 64, 32, stage_idx=0, L=N_STAGES, lp=b["out"].lp_target, stats_col=5,
                                risk_adjustment=True, n_obs=100, p_hist=p_hist, Cc=Cc, lam=1.0)

    for i in range(args.warmup):
        step(i)
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    tokens = sum(int(bufs[(args.warmup + i) % nbuf]["out"].n_acc.sum().item()) + B for i in range(args.steps))
    # kernel-only: back-to-back fused calls (two GEMM launches + the merge launch) between two events
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    runs = []
    for r in range(4):                                     # run 0 settles clocks / power state and is dropped
        e0.record()
        for i in range(30):
            b = bufs[i % nbuf]
            ver(b["h"], b["tok"], b["lp_d"], b["u"], out=b["out"])
        e1.record()
        torch.cuda.synchronize()
        if r:
            runs.append(e0.elapsed_time(e1) / 30)
    kern_ms = sum(runs) / len(runs)
    stats = torch.tensor([elapsed, float(tokens)], dtype=torch.float64, device=red_dev)
    if world > 1:
        tmax = stats[:1].clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = stats[1:].clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        elapsed, tokens = float(tmax.item()), float(tsum.item())
    if rank == 0:
        flops = 2.0 * M * D * V
        hbm = V * D * 2 + M * D * 2
        out = {
            "metric": "verified_tokens_per_s", "value": tokens / elapsed, "unit": "tokens/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"{args.workload} from hidden states: {desc.split(';')[0]}; step = asd_lm_head_verify "
                                   f"({args.lm_head} lm_head, D={D}, packed tile-major: bf16 MFMA GEMM + log-sum-exp + accept, no logits in HBM)"
                                   + ("" if args.verify_only else " + asd_predictor_stop"),
                       "batch_per_gpu": B, "draft_len": K, "vocab": V, "hidden": D, "accumulate": "f32 MFMA (epilogue f64)",
                       "rotating_buffers": nbuf, "launch_mode": "eager",
                       "parallelism": f"batch-parallel replicas x{world}" if world > 1 else "single GPU"},
            "roofline": {"bound": "mfma", "achieved": flops / (kern_ms * 1e-3) / 1e12, "peak": MFMA_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": flops / (kern_ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, "traffic": None,
                         "kernel": "asd::k_lm_head_tile<4> + <2> + k_accept_from_blocks (lm_head_verify.hip)",
                         "algorithmic_flops": flops, "algorithmic_bytes": hbm,
                         "hbm_GBs": hbm / (kern_ms * 1e-3) / 1e9, "hbm_frac": hbm / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "kernel_ms_mean": kern_ms, "kernel_ms_runs": runs,
                         "timing": "HIP events around 3 runs of 30 back-to-back asd_lm_head_verify calls (after one settling run)"},
        }
        if world == 1 and not args.no_cpu_baseline:
            from oracle import oracle as O
            vs = 8192                                           # bounded sample: the first 8192 vocabulary columns
            b = bufs[0]
            hb = b["h"].view(torch.int16).cpu().numpy().view(np.uint16)
            wb = w[:vs].contiguous().view(torch.int16).cpu().numpy().view(np.uint16)
            tk = np.minimum(b["tok"].cpu().numpy(), vs - 1)
            from threadpoolctl import threadpool_limits
            cores = max(1, min(len(os.sched_getaffinity(0)), 16))   # the GPU box's CPU share for one GPU
            t1 = time.perf_counter()
            n = 0
            with threadpool_limits(limits=cores):
                while time.perf_counter() - t1 < min(args.cpu_budget_s, 10.0):
                    O.lm_head_verify(hb, wb, tk, b["lp_d"].cpu().numpy(), b["u"].cpu().numpy(), B, K)
                    n += 1
            per_step = (time.perf_counter() - t1) / n * (V / vs)
            out["cpu_baseline"] = {"value": M / per_step, "unit": "tokens/s (rows scored per second)",
                                   "cores": cores, "kind": "port",
                                   "sample": f"oracle.lm_head_verify (numpy f64 GEMM + f64 accept rule) on {vs} of {V} "
                                             f"vocabulary columns, {n} passes, scaled by V/{vs}"}
        _emit(out)
    if world > 1:
        dist.destroy_process_group()


def main_tiers(args):
    """`--placement tiers`: the timed step is ONE step of the three-tier loop (draft K tokens, tier-1 verify, stop rule,
    escalated blocks re-verified by tier 2, residual / bonus draw, per-sequence commit) with the tiers placed over the
    ranks; value = committed tokens per second of the whole job.  The roofline object is still the verify kernel's,
    measured on rank 0 on a resident c3-shaped buffer."""
    import numpy as np
    import torch
    import torch.distributed as dist

    from asd_amd import kernels as Kmod

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if os.environ.get("ASD_BENCH_ONE_DEVICE") == "1" else int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} does not match the launcher's WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(args.dist_backend, **({"device_id": device} if args.dist_backend == "nccl" else {}))
    B, K, V, desc = WORKLOADS[args.workload]
    B = args.loop_batch
    steps = args.steps if args.steps != 1000 else 8
    warmup = args.warmup if args.warmup != 300 else 2
    sharded = args.placement == "sharded-target"
    if sharded:
        if world == 1:                                  # the sharded head talks to a process group even when it is alone
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                port = sk.getsockname()[1]
            dist.init_process_group(args.dist_backend, init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                                    **({"device_id": device} if args.dist_backend == "nccl" else {}))
        if B == 32:
            B = 16                                      # configs[4]: batch 128 over 8 GPUs
        rec = sharded_target_loop(torch, dist, device, rank, world, args.tier_shapes.split(","), B, K, 32, warmup, steps)
    else:
        rec = hierarchy_loop(torch, dist, device, rank, world, args.tier_shapes.split(","), B, K, 32, warmup, steps,
                             lam=args.lam, target_stop_rate=args.stop_rate)
    if rank == 0:
        torch.cuda.empty_cache()
        nbuf = 3
        ws, bufs = build_inputs(torch, Kmod, B, K, V, nbuf, device, seed=1234)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

        def verify(buf):
            Kmod.verify_accept(buf["logits"], buf["tok"], buf["lp_d"], buf["u"], ws, buf["out"])
        for i in range(600):
            verify(bufs[i % nbuf])
        torch.cuda.synchronize()
        e0.record()
        for i in range(200):
            verify(bufs[i % nbuf])
        e1.record()
        torch.cuda.synchronize()
        kern_ms = e0.elapsed_time(e1) / 200
        nbytes = algorithmic_bytes(B, K, V)
        out = {
            "metric": "verified_tokens_per_s", "value": rec["verified_tokens_per_s"], "unit": "tokens/s", "n_gpus": world,
            "steps": steps, "warmup": warmup, "ms_per_step": rec["ms_per_step"], "higher_is_better": True,
            "scaling": "weak" if sharded else "strong", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": (f"sharded-target: token-level loop, replicated {rec['tiers'][0]} drafts + {rec['tiers'][-1]} target "
                                    f"vocab-sharded over {world} rank(s), batch {B} per rank" if sharded else
                                    f"tiers: token-level loop, {'/'.join(rec['tiers'])} placed over {world} rank(s), batch {B}")
                                   + f", draft_len {K}, vocab {V}; step = draft K tokens (asd_draft_sample) + tier verify "
                                   "(asd_verify_accept / asd_lm_head_verify / asd_lm_head_partial) + asd_predictor_stop + escalation + "
                                   "asd_residual_sample_ex + asd_commit_step; synthetic random-weight models",
                       "batch": B, "draft_len": K, "vocab": V, "parallelism": f"{args.placement} over {world} rank(s)",
                       "placement": rec["placement"]},
            "roofline": {"bound": "hbm", "achieved": nbytes / (kern_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": nbytes / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                         "kernel": "asd::k_verify (verify_accept.hip)", "algorithmic_bytes": nbytes, "kernel_ms_mean": kern_ms,
                         "timing": "HIP events around 200 back-to-back verify launches on rank 0 after the loop (3 rotating buffers)"},
            "loop": rec,
        }
        _emit(out)
    if dist.is_initialized():
        dist.destroy_process_group()


_RESULT_FD = None


def _quiet_rccl_banner():
    """The contract is ONE JSON line on stdout.  RCCL writes its version banner (this image exports NCCL_DEBUG=VERSION)
    and every NCCL WARN (some boxes: 'Missing "iommu=pt"', 'Could not read node #') to the process's STDOUT at
    communicator creation.  So file descriptor 1 is pointed at stderr for the whole run -- native libraries included --
    and the result line goes to a private duplicate of the original stdout (_emit)."""
    global _RESULT_FD
    if os.environ.get("NCCL_DEBUG", "").upper() in ("", "VERSION"):
        os.environ["NCCL_DEBUG"] = "WARN"
    if _RESULT_FD is None:
        sys.stdout.flush()
        _RESULT_FD = os.dup(1)
        os.dup2(2, 1)


def _emit(obj):
    """The one result line, to the real stdout."""
    line = (json.dumps(obj) + "\n").encode()
    if _RESULT_FD is None:
        sys.stdout.write(line.decode())
        sys.stdout.flush()
    else:
        sys.stdout.flush()
        os.write(_RESULT_FD, line)


def sharded_verify_step(torch, dist, device, rank, world, B, K, V, D=8192, reps=30, group=None):
    """The one exchange step of a vocab-sharded target over the ranks of this job (BASELINE configs[4]'s hot path
    without the models): every rank reduces its [V/N, D] slice of a 72B-shape lm_head to (m2, s, g) triples
    (asd_lm_head_partial, bf16 MFMA, no logits), ONE all-gather of [B,K,3] floats over RCCL, asd_accept_from_partials.
    Returns the per-step time (max over ranks) and the bytes a rank sends."""
    from asd_amd.distributed import VocabShardedVerifier
    ver = VocabShardedVerifier(V, group=group)     # `group`: the data-path (RCCL) group; control stays on the default group
    g = torch.Generator(device=device).manual_seed(77)
    hid = torch.randn((B, K, D), generator=g, device=device).to(torch.bfloat16)
    gw = torch.Generator(device=device).manual_seed(1000 + rank)
    w = (torch.randn((ver.v1 - ver.v0, D), generator=gw, device=device) * (3.0 / D ** 0.5)).to(torch.bfloat16)
    tok = torch.randint(0, V, (B, K), generator=g, device=device, dtype=torch.int32)
    lp_d = -torch.rand((B, K), generator=g, device=device) * 2
    u = torch.rand((B, K), generator=g, device=device)
    for _ in range(5):
        out = ver.verify_hidden(hid, w, tok, lp_d, u)
    torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = ver.verify_hidden(hid, w, tok, lp_d, u)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    red = torch.device("cpu") if dist.get_backend() == "gloo" else device
    t = torch.tensor([dt], dtype=torch.float64, device=red)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    n_acc = out[2]
    data_backend = dist.get_backend(group) if group is not None else dist.get_backend()
    return {"what": "vocab-sharded verify step: asd_lm_head_partial on a [V/N, 8192] lm_head shard + all-gather [B,K,3] f32 "
                    "+ asd_accept_from_partials", "ranks": world, "batch": B, "draft_len": K, "vocab": V, "hidden": D,
            "us_per_step": 1e6 * float(t.item()), "bytes_sent_per_rank_per_step": B * K * 12 * (world - 1),
            "tokens_per_step": int(n_acc.sum().item()) + B, "backend": data_backend,
            "control_backend": dist.get_backend()}


def _spawn_ranks(n, argv, timeout_s):
    """`bench.py --gpus N` started as a plain process (no torch.distributed.run, WORLD_SIZE unset): THIS process never
    touches the GPU -- it has not imported torch -- and starts the N ranks as children (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* in their environment, rendezvous on 127.0.0.1), relays rank 0's one JSON line to its own stdout and exits
    non-zero if any rank failed.  Children are started with subprocess (fork + exec of a process that holds no GPU
    state) and ended by their exact PIDs; the torch.distributed.run path is unchanged (WORLD_SIZE set => no spawn)."""
    import socket
    import subprocess
    import threading

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs, lines = [], []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), ASD_BENCH_SPAWNED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL across processes needs it on this driver
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=sys.stderr))

    def pump():
        for raw in procs[0].stdout:
            lines.append(raw.decode(errors="replace").rstrip("\n"))
    th = threading.Thread(target=pump, daemon=True)
    th.start()
    deadline = time.monotonic() + timeout_s
    failed = None
    while True:
        codes = [p.poll() for p in procs]
        bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            failed = f"rank {bad[0][0]} exited with code {bad[0][1]}"
            break
        if all(c == 0 for c in codes):
            break
        if time.monotonic() > deadline:
            failed = f"ranks still running after {timeout_s:.0f} s"
            break
        time.sleep(0.05)
    if failed:
        for p in procs:                       # exactly the processes started above
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=20)
            except subprocess.TimeoutExpired:
                p.kill()
    th.join(timeout=10)
    result = None
    for ln in lines:
        try:
            obj = json.loads(ln)
        except ValueError:
            print(ln, file=sys.stderr)
            continue
        if isinstance(obj, dict) and "metric" in obj:
            result = ln
        else:
            print(ln, file=sys.stderr)
    if result is not None:                    # the headline is relayed even when the job failed afterwards (a hang in the data-path
        sys.stdout.write(result + "\n")       # phase: rank 0 printed the line with the error recorded, then every rank left non-zero)
        sys.stdout.flush()
    if failed:
        print(f"[bench] {failed}", file=sys.stderr)
        return 1
    if result is None:
        print("[bench] rank 0 printed no result line", file=sys.stderr)
        return 1
    return 0


# Which bounded `loop` sub-record the DEFAULT multi-GPU line carries (BASELINE configs[3] / [4]; the reference's placement is
# configs/qwen3_models.yaml:5-53: 7B [0], 32B [1], 72B tensor-parallel over the rest).
LOOP_KEYS = ("kind", "placement", "rccl_ranks", "backend", "verified_tokens_per_s", "ms_per_step", "steps", "roofline", "bytes_sent")


def multi_gpu_plan(world):
    """N = 2 ... 7: the three-tier stop-or-escalate loop with the tiers placed over the ranks (N = 2: {7B + 32B | 72B};
    N >= 4: 7B | 32B | 72B vocab-sharded over up to four ranks), small messages point-to-point.  N >= 8: replicated 7B drafts +
    the 72B target sharded over ALL ranks (work along the batch, lm_head along the vocabulary), batch 16 per rank = 128 at N = 8."""
    if world >= 8:
        return {"kind": "sharded-target", "placement": f"replicated 7b drafts + 72b target sharded over {world} ranks, batch {16 * world}",
                "batch_per_rank": 16}
    tiers = {2: [[0], [1]], 3: [[1], [2]]}.get(world, [[1], list(range(2, min(world, 6)))])
    return {"kind": "tiers", "placement": {"draft": 0, "tiers": tiers, "ranks": world}, "batch": 32}


def main_dry(args):
    """`--dry-run`: the launcher, the rendezvous, the barriers and the max-over-ranks reduction of the N-rank job with NO
    kernel and no GPU (gloo on host tensors).  The line says so (`value` null): it is a test of the plumbing that the CPU
    suite can run (tests/test_bench_launcher.py), never a measurement."""
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
        dist.barrier()
    if args.dry_fail_rank == rank:
        raise SystemExit(3)                   # test hook: a rank that dies must fail the whole job
    t0 = time.perf_counter()
    for _ in range(args.steps):
        pass
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        dist.barrier()
    out = None
    if rank == 0:
        B, K, V, desc = WORKLOADS[args.workload]
        out = {"metric": "verified_tokens_per_s", "value": None, "unit": "tokens/s", "n_gpus": world, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True, "scaling": args.scaling,
               "vs_baseline": None, "dtype": "bf16", "data": "none (dry run of the launcher: no kernel ran)",
               "dry_run": True, "config": {"workload": f"{args.workload}: {desc}", "parallelism": f"{world} rank(s), gloo, no GPU",
                                           "spawned_by_bench": os.environ.get("ASD_BENCH_SPAWNED") == "1"}}
    if world > 1 and not args.no_loop:
        # the data-path phase of the default multi-GPU line (sharded_verify + the bounded loop), as PLUMBING: the same watchdog,
        # the same sub-record keys (values null), one collective standing in for the exchanges
        plan = multi_gpu_plan(world)
        state = {"out": out, "phase": "loop"}
        wd = _PhaseGuard(args.multi_gpu_timeout, rank, world, state, _job_store(dist))
        if args.dry_stall_rank == rank:
            time.sleep(3600)                  # test hook: a rank that hangs in the data path (the others block in the collective)
        rec = dict({k: None for k in LOOP_KEYS}, kind=plan["kind"], placement=plan["placement"], rccl_ranks=world,
                   backend="none (dry run)")
        try:
            if args.dry_raise_rank in (rank, world):      # test hook: this rank's (`world`: every rank's) sub-record raises
                raise RuntimeError("dry-run failure injected")
            dist.all_reduce(torch.zeros(1))
        except Exception as e:  # noqa: BLE001
            rec = {"kind": plan["kind"], "error": f"{type(e).__name__}: {e}"}
            wd.failed(rec["error"])
        wd.cancel()
        if out is not None:
            out["sharded_verify"] = {"ranks": world, "backend": "none (dry run)"}
            out["loop"] = rec
    if rank == 0:
        _emit(out)
    if world > 1:
        dist.destroy_process_group()


class _PhaseGuard:
    """The data-path phase of a multi-GPU run is bounded and a failure on ONE rank is not left to the time bound.

    * Time bound: if a collective hangs, rank 0 prints the headline -- measured before, over gloo -- with the error recorded in the
      sub-record that was running, and EVERY rank leaves with a non-zero code: a process that touched the GPU and then hung is a
      failed run whatever it printed (and must not linger on the device).
    * A rank whose sub-record RAISED reports it through the job's rendezvous store (`failed`).  If every rank raised (a symmetric
      failure: nobody is blocked) the job goes on and the error is recorded in the sub-record, exit code 0.  If only some did, the
      others sit in an exchange that will never complete: every rank's guard sees the report within a second, rank 0 prints the
      headline with that rank's error and every rank leaves non-zero -- seconds, not the time bound."""

    GRACE_S = 5.0

    def __init__(self, seconds, rank, world, state, store=None):
        import threading
        self.seconds, self.rank, self.world, self.state, self.store = seconds, rank, world, state, store
        self._stop = threading.Event()
        # the time bound is a timer of its own: it must fire even while the report poller waits on the store
        self._timer = threading.Timer(seconds, lambda: self._bail(f"timed out after {self.seconds:.0f} s"))
        self._timer.daemon = True
        self._timer.start()
        self._t = threading.Thread(target=self._run, daemon=True)
        self._t.start()

    def _key(self):
        return f"asd_bench_failed/{self.state.get('phase', 'loop')}"

    def _count(self):
        if self.store is None:
            return 0
        try:
            return int(self.store.add(self._key() + "/n", 0))
        except Exception:  # noqa: BLE001  (the store lives on rank 0: gone when rank 0 is gone -- the time bound remains)
            return 0

    def _bail(self, why):
        out, phase = self.state.get("out"), self.state.get("phase", "loop")
        if self.rank == 0 and out is not None:
            out[phase] = {"error": why}
            _emit(out)
        os._exit(3)

    def _run(self):
        seen = None
        while self.store is not None and not self._stop.wait(0.5):
            n = self._count()
            if 0 < n < self.world:
                seen = seen or time.monotonic()
                if time.monotonic() - seen > self.GRACE_S:
                    try:
                        why = self.store.get(self._key() + "/msg").decode(errors="replace")
                    except Exception:  # noqa: BLE001
                        why = "a rank failed"
                    self._bail(f"{why} ({n} of {self.world} ranks failed; the others were ended)")

    def failed(self, msg):
        """This rank's sub-record raised.  Returns when EVERY rank reported a failure of this phase (nobody is blocked); otherwise
        the guard ends the process."""
        if self.store is None:
            return
        try:
            self.store.set(self._key() + "/msg", f"rank {self.rank}: {msg}")
            self.store.add(self._key() + "/n", 1)
        except Exception:  # noqa: BLE001
            return
        t0 = time.monotonic()
        while self._count() < self.world and time.monotonic() - t0 < self.GRACE_S + 30.0:
            time.sleep(0.2)

    def cancel(self):
        self._stop.set()
        self._timer.cancel()


def _job_store(dist):
    """A client connection OF ITS OWN to the job's rendezvous store (MASTER_ADDR:MASTER_PORT: rank 0's, or the launcher agent's):
    the process group's client serialises its operations, and a main thread waiting in a rendezvous would hold the guard's poll."""
    try:
        from datetime import timedelta
        return dist.TCPStore(os.environ["MASTER_ADDR"], int(os.environ["MASTER_PORT"]), None, False,
                             timeout=timedelta(seconds=20), wait_for_workers=False)
    except Exception as e:  # noqa: BLE001  (no store: the time bound alone guards the phase)
        print(f"[bench] no store connection for failure reports ({type(e).__name__}: {e})", file=sys.stderr)
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=300)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="c3")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-workloads", action="store_true", help="skip the kernel-only c2/c5 side measurements")
    ap.add_argument("--cpu-budget-s", type=float, default=12.0)
    ap.add_argument("--splits", type=int, default=0)
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--unroll", type=int, default=0)
    ap.add_argument("--nontemporal", type=int, default=-1)
    ap.add_argument("--verify-only", action="store_true", help="skip the predictor/stop epilogue launch")
    ap.add_argument("--two-launch", action="store_true",
                    help="the step as two launches: asd_verify_accept_ex + asd_predictor_stop (the default until round 2).  The "
                         "default step is ONE launch: asd_verify_accept_fused_ex runs the epilogue inside the verify kernel, by "
                         "the wave that completes each sequence (N1 second form)")
    ap.add_argument("--fused", action="store_true", help=argparse.SUPPRESS)      # the default since round 3; kept for old scripts
    ap.add_argument("--mode", choices=["graph", "eager", "overlap"], default="graph",
                    help="graph: runs of steps captured in one hipGraph on one stream (default; falls back to eager if "
                         "capture fails); eager: plain launches; overlap: epilogue forked to a side stream inside the "
                         "graph (measured SLOWER on this stack: cross-queue edges cost more than the 5 us they hide)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak: every rank verifies a full batch of the workload; strong: the workload's batch is "
                         "split over the ranks (B/rank = B/world)")
    ap.add_argument("--dist-backend", choices=["nccl", "gloo"], default="nccl",
                    help="gloo + ASD_BENCH_ONE_DEVICE=1 rehearses the N>1 control flow with every rank on cuda:0")
    ap.add_argument("--lm-head", choices=sorted(LM_HEADS), default=None,
                    help="start the step from hidden states: asd_lm_head_verify with this Qwen2.5 lm_head size (N2)")
    ap.add_argument("--placement", choices=["replicas", "tiers", "sharded-target"], default="replicas",
                    help="replicas (default): every rank runs the kernel step on its own batch (the headline metric, weak "
                         "scaling).  tiers: the token-level loop with the 7B / 32B / 72B hierarchy placed over the ranks "
                         "(BASELINE configs[3]; 1 rank: all on one GPU, 2: {7B+32B | 72B}, >= 4: 7B | 32B | 72B with a "
                         "vocab-sharded lm_head), stop rule live, small messages point-to-point over RCCL.  sharded-target: "
                         "BASELINE configs[4] -- every rank drafts --loop-batch sequences with its own 7B, the 72B target's "
                         "lm_head is vocab-sharded over all ranks ([B,K,3] all-gather per step); weak scaling")
    ap.add_argument("--tier-shapes", default="7b,32b,72b", help="Qwen2.5 shapes of the tiers (or tiny,tiny,tiny)")
    ap.add_argument("--loop-batch", type=int, default=32)
    ap.add_argument("--loop-steps", type=int, default=12, help="steps of the bounded `loop` sub-record of the default run")
    ap.add_argument("--no-loop", action="store_true", help="skip the bounded 7B/32B/72B loop sub-record")
    ap.add_argument("--lam", type=float, default=None, help="lambda of the loop (default: calibrated to --stop-rate)")
    ap.add_argument("--stop-rate", type=float, default=0.66, help="target share of blocks whose tier-1 verdict is final "
                    "(the reference reports 66.2 %% of requests served by its first stage, README.md:91-94)")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher / rendezvous / reduction plumbing only: no kernel, no GPU, `value` null (CPU test of --gpus N)")
    ap.add_argument("--dry-fail-rank", type=int, default=-1, help=argparse.SUPPRESS)
    ap.add_argument("--dry-stall-rank", type=int, default=-1, help=argparse.SUPPRESS)
    ap.add_argument("--dry-raise-rank", type=int, default=-1, help=argparse.SUPPRESS)
    ap.add_argument("--multi-gpu-timeout", type=float, default=600.0,
                    help="N > 1: seconds the data-path phase (sharded_verify + the bounded loop sub-record) may take before the "
                         "headline is printed with the error recorded and every rank exits non-zero")
    ap.add_argument("--launch-timeout", type=float, default=3000.0,
                    help="--gpus N started without torch.distributed.run: seconds after which the spawned ranks are ended")
    args = ap.parse_args()
    args.fused = not (args.two_launch or args.verify_only)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: become the launcher BEFORE anything initialises the GPU (torch is not imported yet)
        sys.exit(_spawn_ranks(args.gpus, sys.argv[1:], args.launch_timeout))
    _quiet_rccl_banner()
    if args.dry_run:
        return main_dry(args)
    if args.lm_head:
        return main_lm_head(args)
    if args.placement in ("tiers", "sharded-target"):
        return main_tiers(args)

    import numpy as np
    import torch
    import torch.distributed as dist

    from asd_amd import kernels as Kmod

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} does not match the launcher's WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    if os.environ.get("ASD_BENCH_ONE_DEVICE") == "1":
        local_rank = 0                                   # rehearsal on a 1-GPU box (use with --dist-backend gloo)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    distributed = world > 1
    # Control (barriers, the max-over-ranks of the timings) runs over gloo on host tensors: batch-parallel replicas have NO
    # data-path collective, so the headline must not depend on the state of RCCL / xGMI.  RCCL serves the one exchange step
    # there is -- the all-gather of the vocab-sharded verify sub-record -- through its own group.
    red_dev = torch.device("cpu")
    data_group = None
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
        if args.dist_backend == "nccl":
            try:
                data_group = dist.new_group(backend="nccl")
            except Exception as e:  # noqa: BLE001
                print(f"[bench] RCCL group not created ({type(e).__name__}: {e}); sharded_verify will be skipped", file=sys.stderr)

    B, K, V, desc = WORKLOADS[args.workload]
    if args.scaling == "strong":
        if B % world:
            raise SystemExit(f"--scaling strong needs the batch ({B}) to divide by the ranks ({world})")
        B //= world
    bytes_per_launch = algorithmic_bytes(B, K, V)
    nbuf = max(3, math.ceil(640e6 / (B * K * V * 2)))
    ws, bufs = build_inputs(torch, Kmod, B, K, V, nbuf, device, seed=1234 + rank)
    weights = predictor_weights(np)
    packed = Kmod.pack_mlp_weights(*weights, device=device)
    feat_np = (np.random.default_rng(7).standard_normal((B, 64)) * 0.3).astype(np.float32)
    feat = torch.from_numpy(feat_np).to(device)
    Cc = torch.tensor(STAGE_COSTS, dtype=torch.float64, device=device)
    p_hist = torch.ones((B, N_STAGES), dtype=torch.float64, device=device)
    lib = Kmod._lib()
    import ctypes
    from asd_amd._binding import verify_options
    opt = verify_options(1.0, args.splits, args.threads, args.unroll, args.nontemporal)
    opt_ptr = ctypes.addressof(opt)

    # pre-bound launch closures: no allocation, no Python-side tensor work inside the timed loop
    score = torch.empty((B,), dtype=torch.float32, device=device)
    k_star = torch.empty((B,), dtype=torch.int32, device=device)
    stop = torch.empty((B,), dtype=torch.uint8, device=device)
    n_valid = torch.full((B,), K, dtype=torch.int32, device=device)

    def verify(buf, stream=None):
        o = buf["out"]
        st = torch.cuda.current_stream().cuda_stream if stream is None else stream
        rc = lib.asd_verify_accept_ex(buf["logits"].data_ptr(), 1, V, buf["tok"].data_ptr(), buf["lp_d"].data_ptr(),
                                      buf["u"].data_ptr(), B, K, V, o.lp_target.data_ptr(), o.accept.data_ptr(),
                                      o.n_acc.data_ptr(), o.accept_bits.data_ptr(), ws.buf.data_ptr(), ws.bytes,
                                      opt_ptr, st)
        if rc:
            raise RuntimeError(f"asd_verify_accept_ex rc={rc}")

    def epilogue(buf, stream=None):
        o = buf["out"]
        st = torch.cuda.current_stream().cuda_stream if stream is None else stream
        rc = lib.asd_predictor_stop(o.lp_target.data_ptr(), K, n_valid.data_ptr(), K, feat.data_ptr(), 64, 5,
                                    packed.data_ptr(), 64, 32, 1, 100, 1.0, 1.0, p_hist.data_ptr(), Cc.data_ptr(), 1.0,
                                    N_STAGES, 0, 0, None, B, score.data_ptr(), k_star.data_ptr(), stop.data_ptr(),
                                    None, None, st)
        if rc:
            raise RuntimeError(f"asd_predictor_stop rc={rc}")

    def fused_step(buf, stream=None):
        o = buf["out"]
        st = torch.cuda.current_stream().cuda_stream if stream is None else stream
        rc = lib.asd_verify_accept_fused(buf["logits"].data_ptr(), 1, V, buf["tok"].data_ptr(), buf["lp_d"].data_ptr(),
                                         buf["u"].data_ptr(), B, K, V, o.lp_target.data_ptr(), o.accept.data_ptr(),
                                         o.n_acc.data_ptr(), o.accept_bits.data_ptr(), ws.buf.data_ptr(), ws.bytes,
                                         feat.data_ptr(), 64, 5, packed.data_ptr(), 64, 32, 1, 100, 1.0, 1.0,
                                         p_hist.data_ptr(), Cc.data_ptr(), 1.0, N_STAGES, 0, 0, None, score.data_ptr(),
                                         k_star.data_ptr(), stop.data_ptr(), None, None, st)
        if rc:
            raise RuntimeError(f"asd_verify_accept_fused rc={rc}")

    def step(i):
        buf = bufs[i % nbuf]
        if args.fused:
            fused_step(buf)
            return
        verify(buf)
        if not args.verify_only:
            epilogue(buf)

    def barrier():
        if distributed:
            dist.barrier()

    def capture(first, count, overlap):
        """`count` consecutive steps starting at step index `first` as ONE hipGraph.  overlap: the epilogue
        of step i is forked to a side stream (edge: after verify i) and joined at the end of the graph, so
        it runs under verify i+1.  The buffers a step touches are distinct for nbuf consecutive steps."""
        g = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream(device=device)
        with torch.cuda.graph(g):
            main = torch.cuda.current_stream()
            for j in range(count):
                buf = bufs[(first + j) % nbuf]
                if args.fused:
                    fused_step(buf, main.cuda_stream)
                    continue
                verify(buf, main.cuda_stream)
                if args.verify_only:
                    continue
                if overlap:
                    ev = torch.cuda.Event()
                    ev.record(main)
                    side.wait_event(ev)
                    epilogue(buf, side.cuda_stream)
                else:
                    epilogue(buf, main.cuda_stream)
            if overlap and not args.verify_only:
                ev = torch.cuda.Event()
                ev.record(side)
                main.wait_event(ev)
        return g

    # The driver runs short jobs (--steps 20 --warmup 5: a 0.4 ms timed region) on a fresh box: W steps do not bring the GPU
    # out of its idle clock / memory power state (the first ~1000 launches after an idle gap run 5-15 % slower), so the same
    # step is launched untimed for ~120 ms first (20 ms left a fresh box 2 % short of its steady state).  The W warm-up steps and the K timed steps follow unchanged.
    SETTLE_STEPS = 6000
    for i in range(SETTLE_STEPS):
        step(i)
    torch.cuda.synchronize()
    for i in range(args.warmup):
        step(i)
    # steps per graph: a divisor of --steps (<= 64) so that replays cover EXACTLY --steps steps (a short job is ONE replay:
    # the host cost of a replay, 10-20 us, is 5 % of a 20-step region).  Graph step j always runs buffer
    # (warmup + j) % nbuf, so the token accounting below holds for every mode
    G = 1
    if args.mode != "eager":
        for cand in range(min(args.steps, 64), 0, -1):
            if args.steps % cand == 0:
                G = cand
                break
    graph = None
    if args.mode != "eager" and G > 1:
        torch.cuda.synchronize()
        try:
            graph = capture(args.warmup, G, overlap=(args.mode == "overlap"))
            graph.replay()                              # one untimed replay: graph upload
            torch.cuda.synchronize()
        except Exception as e:  # noqa: BLE001  (e.g. capture refused under a profiler): same kernels, eager
            print(f"[bench] hipGraph capture failed ({type(e).__name__}: {e}); running eager", file=sys.stderr)
            graph, G = None, 1
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if graph is not None:
        for _ in range(args.steps // G):
            graph.replay()
    else:
        for i in range(args.steps):
            step(args.warmup + i)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0       # this rank's K steps; the job's time is the MAX over ranks (below)
    barrier()                                 # (the closing barrier itself -- a host round trip over gloo -- is not work)

    # verified tokens: outputs per buffer are deterministic, so count them after the timed region
    per_buf = [int(b["out"].n_acc.sum().item()) + B for b in bufs]
    tokens = sum(per_buf[(args.warmup + (i % G)) % nbuf] for i in range(args.steps))
    # kernel-only duration: back-to-back verify launches bracketed by two events, best-of-3 + mean
    reps = max(args.steps, 200)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for i in range(8000):                     # settle (~120 ms): the first launches after an idle gap run 5-15 % slower, and a
                                              # short job (--steps 20) has not run long enough to reach the steady state
        verify(bufs[i % nbuf])                # (clock / memory power state ramp; measured: 18.4 -> 16.0 us over 5 runs of 400)
    torch.cuda.synchronize()
    runs = []
    for _ in range(5):
        barrier()
        e0.record()
        for i in range(reps):
            verify(bufs[i % nbuf])
        e1.record()
        torch.cuda.synchronize()
        runs.append(e0.elapsed_time(e1) / reps)
    kern_mean_ms = sum(runs) / len(runs)
    kern_min_ms = min(runs)
    # the same measurement for the kernel the DEFAULT step launches (the FUSED instantiation: verify + in-kernel epilogue)
    fused_runs = []
    if args.fused:
        for i in range(2000):
            fused_step(bufs[i % nbuf])
        torch.cuda.synchronize()
        for _ in range(5):
            barrier()
            e0.record()
            for i in range(reps):
                fused_step(bufs[i % nbuf])
            e1.record()
            torch.cuda.synchronize()
            fused_runs.append(e0.elapsed_time(e1) / reps)

    # side measurement (never `value`): TWO independent batches in flight -- the same one-launch step on two streams with their own
    # workspaces and outputs, forked and joined inside one hipGraph.  What separates a serial step from the 9.7 us its bytes take at
    # 8 TB/s is mostly the ramp to the first byte and the tail behind the last one; a second batch hides them.
    two_stream = None
    if args.fused and world == 1 and not args.no_other_workloads:
        try:
            ws_b = Kmod.VerifyWorkspace(B, K, V, torch.bfloat16, device)
            score_b, k_star_b, stop_b, p_hist_b = torch.empty_like(score), torch.empty_like(k_star), torch.empty_like(stop), p_hist.clone()

            def fused_step_b(buf, st):
                o = buf["out"]
                rc = lib.asd_verify_accept_fused(buf["logits"].data_ptr(), 1, V, buf["tok"].data_ptr(), buf["lp_d"].data_ptr(),
                                                 buf["u"].data_ptr(), B, K, V, o.lp_target.data_ptr(), o.accept.data_ptr(),
                                                 o.n_acc.data_ptr(), o.accept_bits.data_ptr(), ws_b.buf.data_ptr(), ws_b.bytes,
                                                 feat.data_ptr(), 64, 5, packed.data_ptr(), 64, 32, 1, 100, 1.0, 1.0,
                                                 p_hist_b.data_ptr(), Cc.data_ptr(), 1.0, N_STAGES, 0, 0, None, score_b.data_ptr(),
                                                 k_star_b.data_ptr(), stop_b.data_ptr(), None, None, st)
                if rc:
                    raise RuntimeError(f"asd_verify_accept_fused rc={rc}")
            pairs = 12
            sa, sb = torch.cuda.Stream(device=device), torch.cuda.Stream(device=device)
            torch.cuda.synchronize()
            g2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g2):
                main_s = torch.cuda.current_stream()
                ev = torch.cuda.Event()
                ev.record(main_s)
                sa.wait_event(ev)
                sb.wait_event(ev)
                for j in range(pairs):
                    fused_step(bufs[(2 * j) % nbuf], sa.cuda_stream)
                    fused_step_b(bufs[(2 * j + 1) % nbuf], sb.cuda_stream)
                ea, eb = torch.cuda.Event(), torch.cuda.Event()
                ea.record(sa)
                eb.record(sb)
                main_s.wait_event(ea)
                main_s.wait_event(eb)
            g2.replay()
            torch.cuda.synchronize()
            ts_runs = []
            for _ in range(5):
                e0.record()
                for _ in range(10):
                    g2.replay()
                e1.record()
                torch.cuda.synchronize()
                ts_runs.append(e0.elapsed_time(e1) / (10 * 2 * pairs))
            del g2
            ts_ms = sum(ts_runs) / len(ts_runs)
            two_stream = {"what": "the one-launch step on TWO streams (two independent batches in flight, own workspaces and outputs), "
                                  f"{2 * pairs} steps per hipGraph replay; per-step time = replay time / steps.  Context only: `value` is the serial step",
                          "ms_per_step": ts_ms, "ms_per_step_runs": ts_runs,
                          "achieved_GBs": bytes_per_launch / (ts_ms * 1e-3) / 1e9, "frac_of_8TBs": bytes_per_launch / (ts_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
        except Exception as e:  # noqa: BLE001  (context record: the headline does not depend on it)
            two_stream = {"error": f"{type(e).__name__}: {e}"}
            torch.cuda.synchronize()

    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        tk = torch.tensor([tokens], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tk, op=dist.ReduceOp.SUM)
        tokens = int(tk.item())
        km = torch.tensor([kern_mean_ms], dtype=torch.float64, device=red_dev)
        dist.all_reduce(km, op=dist.ReduceOp.MAX)
        kern_mean_ms = float(km.item())

    cpu_buf = bufs[0]
    # kernel-only figures for the other BASELINE shapes (not bench lines: context for the roofline)
    others = {}
    if rank == 0 and world == 1 and not args.no_other_workloads:
        bufs = None
        torch.cuda.empty_cache()
        for name in sorted(WORKLOADS):
            if name == args.workload:
                continue
            oB, oK, oV, _ = WORKLOADS[name]
            onb = max(3, math.ceil(640e6 / (oB * oK * oV * 2)))
            ows, obufs = build_inputs(torch, Kmod, oB, oK, oV, onb, device, seed=99)

            def overify(buf):
                o = buf["out"]
                return lib.asd_verify_accept(buf["logits"].data_ptr(), 1, oV, buf["tok"].data_ptr(),
                                             buf["lp_d"].data_ptr(), buf["u"].data_ptr(), oB, oK, oV,
                                             o.lp_target.data_ptr(), o.accept.data_ptr(), o.n_acc.data_ptr(),
                                             o.accept_bits.data_ptr(), ows.buf.data_ptr(), ows.bytes,
                                             torch.cuda.current_stream().cuda_stream)
            ofeat = feat[:1].expand(oB, 64).contiguous()
            oph = torch.ones((oB, N_STAGES), dtype=torch.float64, device=device)
            osc = torch.empty((oB,), dtype=torch.float32, device=device)
            oks = torch.empty((oB,), dtype=torch.int32, device=device)
            ost = torch.empty((oB,), dtype=torch.uint8, device=device)

            def ofused(buf):
                o = buf["out"]
                return lib.asd_verify_accept_fused(buf["logits"].data_ptr(), 1, oV, buf["tok"].data_ptr(),
                                                   buf["lp_d"].data_ptr(), buf["u"].data_ptr(), oB, oK, oV,
                                                   o.lp_target.data_ptr(), o.accept.data_ptr(), o.n_acc.data_ptr(),
                                                   o.accept_bits.data_ptr(), ows.buf.data_ptr(), ows.bytes, ofeat.data_ptr(), 64, 5,
                                                   packed.data_ptr(), 64, 32, 1, 100, 1.0, 1.0, oph.data_ptr(), Cc.data_ptr(), 1.0,
                                                   N_STAGES, 0, 0, None, osc.data_ptr(), oks.data_ptr(), ost.data_ptr(), None, None,
                                                   torch.cuda.current_stream().cuda_stream)
            ob = algorithmic_bytes(oB, oK, oV)
            # settle for ~60 ms of launches (clock / memory power state ramp after the allocation gap: a short kernel
            # needs thousands of launches for that, 300 left the B=8 figure 30 % above its steady state)
            for i in range(max(400, int(60e-3 / (ob / 4.0e12)))):
                overify(obufs[i % onb])
            # a B=8 launch (8 us) is shorter than a Python ctypes call: time replays of a hipGraph of the launches
            # (same kernels, same rotating buffers), eager only if capture is refused (e.g. under a profiler)
            per = 24                # 24 launches per graph whatever the buffer count: a replay boundary costs ~10 us,
                                    # 6 % of a graph of three B=128 launches

            def time_launches(fn):
                og = None
                try:
                    torch.cuda.synchronize()
                    og = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(og):
                        for i in range(per):
                            fn(obufs[i % onb])
                    og.replay()
                    torch.cuda.synchronize()
                except Exception:  # noqa: BLE001
                    og = None
                oreps = 10 if og is not None else 200
                oruns = []
                for _ in range(5):
                    e0.record()
                    for i in range(oreps):
                        if og is not None:
                            og.replay()
                        else:
                            fn(obufs[i % onb])
                    e1.record()
                    torch.cuda.synchronize()
                    oruns.append(e0.elapsed_time(e1) / (oreps * (per if og is not None else 1)))
                del og
                return oruns
            oruns = time_launches(overify)
            oms = sum(oruns) / len(oruns)
            fruns = time_launches(ofused)
            fms = sum(fruns) / len(fruns)
            others[name] = {"batch": oB, "draft_len": oK, "vocab": oV, "algorithmic_bytes": ob, "kernel_ms_mean": oms,
                            "kernel_ms_runs": oruns, "launch": "hipGraph replays of back-to-back launches", "achieved_GBs": ob / (oms * 1e-3) / 1e9,
                            "frac_of_8TBs": ob / (oms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                            "step_one_launch_ms_mean": fms, "step_one_launch_ms_runs": fruns,
                            "step_note": "asd_verify_accept_fused: verify + in-kernel predictor / Bayes / DP epilogue, one launch"}
            del obufs, ows
            torch.cuda.empty_cache()

    loop_rec = None
    if world == 1 and not args.no_loop:
        # bounded sub-record: the full token-level loop around the hot path (7B draft / 32B / 72B on this one GPU,
        # stop rule live), so that "verified tokens/s, 7B-draft/72B-target" has driver-run evidence beside the
        # kernel-step rate.  Never part of `value`.
        try:
            bufs = None                       # release the rotating logits buffers
            torch.cuda.empty_cache()
            loop_rec = hierarchy_loop(torch, dist, device, rank, world, args.tier_shapes.split(","), args.loop_batch, K,
                                      32, 2, args.loop_steps, lam=args.lam, target_stop_rate=args.stop_rate)
        except Exception as e:  # noqa: BLE001  (the headline must not depend on the context record)
            loop_rec = {"error": f"{type(e).__name__}: {e}"}
        torch.cuda.empty_cache()

    if rank == 0:
        # roofline: the kernel the TIMED step launches (the FUSED instantiation unless --two-launch / --verify-only); the plain
        # streaming kernel's figures sit beside it as `plain_kernel`
        step_is_fused = bool(fused_runs)
        plain_ms, plain_min_ms, plain_runs = kern_mean_ms, kern_min_ms, runs
        if step_is_fused:
            kern_mean_ms, kern_min_ms, runs = sum(fused_runs) / len(fused_runs), min(fused_runs), fused_runs
        achieved = bytes_per_launch / (kern_mean_ms * 1e-3) / 1e9
        traffic, traffic_src = load_traffic(fused=step_is_fused)
        out = {
            "metric": "verified_tokens_per_s",
            "value": tokens / elapsed,
            "unit": "tokens/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "bf16",
            "data": "synthetic",
            "config": {"workload": f"{args.workload}: {desc}; step = "
                                   + ("asd_verify_accept_fused_ex: ONE launch (verify + accept + in-kernel stats->MLP->Bayes->DP epilogue)" if args.fused else
                                      "asd_verify_accept" + ("" if args.verify_only else
                                                             " + asd_predictor_stop (stats->MLP->Bayes->DP)")),
                       "batch_per_gpu": B, "draft_len": K, "vocab": V, "accumulate": "f32 (epilogue f64)",
                       "rotating_buffers": nbuf, "buffer_MB": round(B * K * V * 2 / 1e6, 2),
                       "launch_mode": args.mode if graph is not None else "eager", "steps_per_graph": G,
                       "settle": f"{SETTLE_STEPS} untimed steps before the {args.warmup} warm-up steps (clock / power-state ramp)",
                       "tiers": "7B-draft / 32B / 72B-target shapes (vocab 152064); logits synthetic",
                       "parallelism": f"batch-parallel replicas x{world}" if world > 1 else "single GPU",
                       "geometry": {"splits": args.splits, "threads": args.threads, "unroll": args.unroll,
                                    "nontemporal": args.nontemporal}},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "traffic": None if traffic is None else traffic.get("hbm_bytes_per_launch"),
                         "traffic_source": traffic_src,
                         "kernel": ("asd::k_verify<..., FUSED = true> (verify_accept.hip): verify + accept + the in-kernel statistics / predictor / "
                                    "Bayes / DP epilogue -- the ONE kernel a timed step launches") if step_is_fused else
                                   "asd::k_verify<..., FUSED = false> (verify_accept.hip): the step's dominant kernel",
                         "algorithmic_bytes": bytes_per_launch,
                         "kernel_ms_mean": kern_mean_ms, "kernel_ms_best_run": kern_min_ms, "kernel_ms_runs": runs,
                         "timing": f"HIP events on the launch stream around 5 runs of {reps} back-to-back launches of that kernel "
                                   "(rotating buffers) right after the timed region; includes inter-kernel gaps",
                         "note": "event PAIRS around single launches add 5-15 us each on this stack (measured in round 1) and are not used"},
        }
        if step_is_fused:
            out["roofline"]["plain_kernel"] = {
                "kernel": "asd::k_verify<..., FUSED = false>: the streaming verify + accept kernel alone (what --two-launch runs as its first launch)",
                "kernel_ms_mean": plain_ms, "kernel_ms_best_run": plain_min_ms, "kernel_ms_runs": plain_runs,
                "achieved": bytes_per_launch / (plain_ms * 1e-3) / 1e9,
                "frac": bytes_per_launch / (plain_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "note": "same algorithmic bytes, same measurement; NOT the kernel `value` was computed from"}
        if two_stream is not None:
            out["roofline"]["two_batches_in_flight"] = two_stream
        if others:
            out["roofline"]["other_workloads_kernel_only"] = others
        if loop_rec is not None:
            out["loop"] = loop_rec
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(np, torch, cpu_buf, B, K, V, weights, feat_np, args.cpu_budget_s)
    else:
        out = None
    if distributed and not args.no_loop:
        # N > 1, default arguments: the data path of the multi-GPU configurations, bounded by ONE watchdog (a hang prints the
        # headline with the error recorded and exits non-zero):
        #   sharded_verify  the exchange step of a vocab-sharded 72B-shape head over this job's ranks (kernels + one all-gather)
        #   loop            BASELINE configs[3] at N = 2 ... 7 (tiers placed over the ranks), configs[4] at N >= 8 (replicated
        #                   drafts + sharded target, batch 16 per rank), real shapes, a few steps
        state = {"out": out, "phase": "sharded_verify"}
        wd = _PhaseGuard(args.multi_gpu_timeout, rank, world, state, _job_store(dist))
        bufs = cpu_buf = None
        torch.cuda.empty_cache()
        try:
            sharded_rec = sharded_verify_step(torch, dist, device, rank, world, B, K, V, group=data_group)
        except Exception as e:  # noqa: BLE001
            sharded_rec = {"error": f"{type(e).__name__}: {e}"}
            wd.failed(sharded_rec["error"])    # returns if every rank failed alike; a one-sided failure ends the job (non-zero)
        if out is not None:
            out["sharded_verify"] = sharded_rec
        state["phase"] = "loop"
        plan = multi_gpu_plan(world)
        torch.cuda.empty_cache()
        try:
            if plan["kind"] == "sharded-target":
                shapes_ = args.tier_shapes.split(",")
                rec = sharded_target_loop(torch, dist, device, rank, world, [shapes_[0], shapes_[-1]], plan["batch_per_rank"], K, 32, 2,
                                          args.loop_steps, group=data_group)
            else:
                rec = hierarchy_loop(torch, dist, device, rank, world, args.tier_shapes.split(","), args.loop_batch, K, 32, 2,
                                     args.loop_steps, lam=args.lam, target_stop_rate=args.stop_rate, group=data_group)
            if rec is not None:
                rec["kind"] = plan["kind"]
        except Exception as e:  # noqa: BLE001  (the headline must not depend on the context record)
            rec = {"kind": plan["kind"], "error": f"{type(e).__name__}: {e}"}
            wd.failed(rec["error"])
        wd.cancel()
        if out is not None:
            out["loop"] = rec
    if rank == 0:
        _emit(out)
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
