"""Three-tier token-level loop with the optimal-stopping rule LIVE: draft -> verify -> stop-or-escalate
(BASELINE configs[3] / [4]; the reference's stop-or-continue loop is src/serving/pipeline.py:248-266).

The reference runs its stages as a cascade of whole answers: stage i generates, the quality predictor turns the
stage's log-probs into an acceptance probability p_i, `bayesian_adjustment` shrinks it, `optimal_stopping_rule`
decides whether stage i's output is final (pipeline.py:225-261).  Here the same decision runs per drafted BLOCK:

  tier 0   (7B)   drafts K tokens per sequence (asd_draft_sample: temperature + top-p, log q(tok)); its "output"
                  is the block itself, judged from its own log-probs lp_d  ->  p_0
  tier s>=1       verifies the SAME block for the sequences still active (asd_verify_accept / asd_lm_head_verify):
                  lp_t^(s), accept mask, n_acc^(s); judged from lp_t^(s)  ->  p_s
  stop rule       asd_predictor_stop(stage_idx = s, p_hist carried): p_s = bayes(predictor(stats(lp))),
                  k* = optimal_stopping_rule(p_hist[b, :], C, lambda) over ALL L tiers (tiers not run yet keep the
                  prior 1.0, the last tier is forced);  sequence b STOPS at tier s iff k* <= s, else it ESCALATES:
                  tier s+1 re-verifies the same drafted tokens (its own uniforms), after catching its KV cache up
                  on the tokens committed while it was idle.
  commit          the verdict of the tier a sequence stopped at is final: n_acc accepted tokens + one token drawn by
                  that tier from the residual max(0, p_t - p_d) at the first rejection (p_d = the nucleus-truncated
                  draft distribution) or from its bonus row; asd_commit_step appends per sequence (ragged).
  min_verify_stage = 1 (default): a block is verified at least once (k* = 0 is read as "continue"), so the output
                  is always distributed as SOME verifying tier's distribution.  0 = the reference's pure cascade:
                  a sequence whose k* is 0 commits its K drafted tokens unverified.

Everything above is arithmetic behind `ops` (distributed.HipOps -> libasd_hip.so; the CPU test-suite injects the
oracle, tests/oracle_backend.py).  Model execution (SyntheticLM, per-sequence KV) and message passing are torch
plumbing.  One code path serves every placement: a `Placement` maps the roles "draft", "t1", "t2", ... to ranks;
roles on the same rank hand tensors over directly, roles on different ranks use point-to-point send / recv
(RCCL over xGMI on the GPU box, gloo in the CPU tests) of the SMALL messages only:

  draft -> tier s     tok [B,K] i32, lp_d [B,K] f32, p_0 [B] f64, stop_0 [B]            (2.3 KB at B=32, K=8)
  tier s -> tier s+1  escalate [B] u8 + p_hist[:, :s+1] f64                              (< 1 KB)
  tier s -> draft     active / stop / n_acc [B] i32                                      (384 B)
  draft -> tier s     ONE draft-logits row (storage dtype) + its nucleus threshold per sequence that stopped at
                      tier s WITH a rejection (all-accepted sequences draw from the tier's own bonus row)
  tier s -> draft     drawn [B] i32;   draft -> all: final n_acc / drawn / tier [B] i32

The [B,K,V] target logits never leave the rank that produced them.  A tier whose lm_head is vocab-sharded over a
group of ranks (72B over 2 / 4 GPUs) reduces each shard to (m2, s, g) triples (asd_lm_head_partial), all-gathers
[n,K,3] floats, and all-gathers the shard pieces of the one row per stopping sequence its draw needs.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist

from ..distributed import HipOps, all_gather_any, host_staged, shard_bounds


@dataclass
class HierarchyConfig:
    draft_len: int = 8
    temperature: float = 0.7                 # generate_training_data.py:110-119, pipeline.py:94
    top_p: float = 0.9                       # nucleus of the DRAFT tier (>= 1: off)
    stage_costs: Sequence[float] = (1.0, 4.5, 10.0)   # BASELINE.md cost units 7B / 32B / 72B
    lambda_value: float = 1.0
    risk_adjustment: bool = True             # pipeline.py:234-238
    risk_alpha: float = 1.0
    risk_beta: float = 1.0
    n_obs: int = 100
    stats_col: int = 5                       # extract_features columns [5:10] (generate_training_data.py:166-175)
    min_verify_stage: int = 1
    seed: int = 0


def prompt_features(prompt_ids: torch.Tensor) -> torch.Tensor:
    """A9 (minimal_adaptive_decoder.py:51-68) from token ids: [len/512, distinct/len, words/100, 0 x 61] per sequence;
    with ids only, a "word" is a token.  float32 [B, 64] on the prompt's device."""
    from ..minimal_adaptive_decoder import features_from_token_ids
    rows = [features_from_token_ids(r, " ".join("t" for _ in r)) for r in prompt_ids.cpu().tolist()]
    return torch.from_numpy(np.stack(rows)).to(prompt_ids.device)


# ------------------------------------------------------------------------------------------- messages
@dataclass
class DraftMsg:
    tok: torch.Tensor      # [B,K] i32
    lp_d: torch.Tensor     # [B,K] f32
    p0: torch.Tensor       # [B] f64   stage-0 adjusted probability
    stop0: torch.Tensor    # [B] u8    1: the block is committed unverified (min_verify_stage == 0 only)


@dataclass
class EscMsg:
    escalate: torch.Tensor  # [B] u8
    p_prev: torch.Tensor    # [B, s+1] f64: p_hist columns 0..s of the tiers that already judged the block


@dataclass
class Verdict:
    active: torch.Tensor   # [B] i32  1: this tier verified the sequence this step
    stop: torch.Tensor     # [B] i32  1: ... and its verdict is final
    n_acc: torch.Tensor    # [B] i32  accepted prefix (valid where active)
    # trace only (never cross a link)
    idx: Optional[torch.Tensor] = None
    accept: Optional[torch.Tensor] = None
    k_star: Optional[torch.Tensor] = None
    score: Optional[torch.Tensor] = None
    p_hist: Optional[torch.Tensor] = None
    inputs: Optional[dict] = None


@dataclass
class FinalMsg:
    n_acc: torch.Tensor    # [B] i32
    drawn: torch.Tensor    # [B] i32
    tier: torch.Tensor     # [B] i32  the tier whose verdict was committed


def _timed(role, fn):
    """Run a model pass; with role.events a list, bracket it with HIP events on the current stream (bench.py: the per-tier
    torch time beside the hot-path calls)."""
    if role.events is None or not torch.cuda.is_available():
        return fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    out = fn()
    e1.record()
    role.events.append((e0, e1))
    return out


def _dev_gen(device, seed: int) -> torch.Generator:
    return torch.Generator(device=device).manual_seed(int(seed))


class _SeqState:
    """The per-role replica of the committed token buffer (every role commits the same FinalMsg)."""

    def __init__(self, prompt_ids: torch.Tensor, max_new_tokens: int, K: int):
        dev = prompt_ids.device
        self.B, self.P = prompt_ids.shape
        if self.P < 2:
            raise ValueError("the loop needs a prompt of at least two tokens")
        self.K = K
        self.cap = self.P + max_new_tokens
        self.kv_slots = self.cap + K + 3                     # + the trash slot forward_ragged clamps padding into
        self.tokens = torch.zeros((self.B, self.cap), dtype=torch.int32, device=dev)
        self.tokens[:, :self.P] = prompt_ids.to(torch.int32)
        self.seq_len = torch.full((self.B,), self.P, dtype=torch.int32, device=dev)
        self.n_commit = torch.zeros((self.B,), dtype=torch.int32, device=dev)
        self.steps = 0

    def window(self) -> int:
        """Host-side bound on every position a step can touch (no device read-back)."""
        return min(self.P + self.steps * (self.K + 1) + self.K + 2, self.kv_slots)

    def commit(self, ops, tok: torch.Tensor, final: FinalMsg) -> None:
        ops.commit_step(tok, final.n_acc, final.drawn, self.seq_len, self.tokens, self.n_commit, self.cap)
        self.steps += 1


# ------------------------------------------------------------------------------------------- tier 0
class DraftRole:
    def __init__(self, model, cfg: HierarchyConfig, ops, prompt_ids: torch.Tensor, max_new_tokens: int, predictor,
                 feat: Optional[torch.Tensor] = None, batch_total: Optional[int] = None, batch_offset: int = 0):
        """batch_total / batch_offset: this role drafts sequences [offset, offset + B) of a batch of `batch_total`
        (replicated drafts, BASELINE configs[4]); its uniforms are the matching slice of the full batch's draws, so
        the stream does not depend on how the batch is cut over ranks."""
        self.m, self.cfg, self.ops = model, cfg, ops
        self.batch_total = batch_total if batch_total is not None else prompt_ids.shape[0]
        self.batch_offset = batch_offset
        self.st = _SeqState(prompt_ids, max_new_tokens, cfg.draft_len)
        dev = prompt_ids.device
        B, K = self.st.B, cfg.draft_len
        self.gen = _dev_gen(dev, cfg.seed)
        self.inv_t = float(np.float32(1.0 / cfg.temperature))
        self.L = len(cfg.stage_costs)
        self.costs = torch.tensor(list(cfg.stage_costs), dtype=torch.float64, device=dev)
        self.pred = ops.pack_predictor(predictor, dev)
        self.feat = prompt_features(prompt_ids) if feat is None else feat
        model.reset()
        model.alloc_ragged(B, self.st.kv_slots)
        if self.st.P > 2:
            model.forward_ragged(prompt_ids[:, :self.st.P - 2], torch.zeros((B,), dtype=torch.int64, device=dev), self.st.P)
        self.rows_b = torch.arange(B, device=dev)
        self.fwd_calls = self.fwd_positions = 0            # model passes / sequence-positions computed (loop roofline)
        self.events: Optional[list] = None                 # set to [] to bracket the model passes with HIP events
        self.d_logits: Optional[torch.Tensor] = None       # [B,K,V] of the current block (storage dtype)
        self.thr = torch.empty((B, K), dtype=torch.float32, device=dev)
        self.tok = torch.empty((B, K), dtype=torch.int32, device=dev)
        self.lp_d = torch.empty((B, K), dtype=torch.float32, device=dev)

    @torch.no_grad()
    def propose(self) -> DraftMsg:
        st, cfg, K = self.st, self.cfg, self.cfg.draft_len
        B = st.B
        L = st.seq_len.to(torch.int64)
        w = st.window()
        last2 = torch.stack([st.tokens[self.rows_b, L - 2], st.tokens[self.rows_b, L - 1]], 1).to(torch.int64)
        dl = _timed(self, lambda: self.m.forward_ragged(last2, L - 2, w))[:, -1]
        self.fwd_calls += 1
        self.fwd_positions += 2 * B
        if self.d_logits is None:
            self.d_logits = torch.empty((B, K, dl.shape[-1]), dtype=dl.dtype, device=dl.device)
        for k in range(K):
            self.d_logits[:, k] = dl
            r = torch.rand((self.batch_total,), generator=self.gen, device=dl.device)[self.batch_offset:self.batch_offset + B].contiguous()
            t, lp, thr = self.ops.draft_sample(self.d_logits[:, k], r, self.inv_t, cfg.top_p)
            self.tok[:, k], self.lp_d[:, k], self.thr[:, k] = t, lp, thr
            if k + 1 < K:
                dl = _timed(self, lambda: self.m.forward_ragged(t[:, None].to(torch.int64), L + k, w))[:, -1]
                self.fwd_calls += 1
                self.fwd_positions += B
        # stage 0 of the stop rule: the draft tier judged from its own log-probs
        ph = torch.ones((B, self.L), dtype=torch.float64, device=dl.device)
        _, k0, ph = self.ops.predictor_stop(self.pred, self.lp_d, self.feat, ph, 0, self.costs, cfg.lambda_value,
                                            cfg.risk_adjustment, cfg.n_obs, cfg.risk_alpha, cfg.risk_beta, cfg.stats_col)
        stop0 = (k0 <= 0).to(torch.uint8) if cfg.min_verify_stage <= 0 else torch.zeros((B,), dtype=torch.uint8, device=dl.device)
        if self.L == 1:
            stop0.fill_(1)
        return DraftMsg(self.tok.clone(), self.lp_d.clone(), ph[:, 0].contiguous(), stop0)

    def rows_for(self, v: Verdict) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """The draft-logits rows tier s needs for its residual draws: one per sequence that stopped there with a
        rejection, ascending in b.  Returns (b indices, rows [n,V], thresholds [n])."""
        need = (v.stop == 1) & (v.n_acc < self.cfg.draft_len)
        b = need.nonzero()[:, 0]
        j = v.n_acc.to(torch.int64)[b]
        return b, self.d_logits[b, j].contiguous(), self.thr[b, j].contiguous()

    def assemble(self, dm: DraftMsg, verdicts: List[Tuple[int, Verdict, torch.Tensor]]) -> FinalMsg:
        """verdicts: (stage_idx, Verdict, drawn [B]) of every tier that ran this step."""
        K = self.cfg.draft_len
        n_acc = torch.full_like(self.st.seq_len, K - 1)
        drawn = dm.tok[:, K - 1].clone()
        tier = torch.zeros_like(self.st.seq_len)
        for s, v, d in verdicts:
            m = v.stop == 1
            n_acc = torch.where(m, v.n_acc, n_acc)
            drawn = torch.where(m, d, drawn)
            tier = torch.where(m, torch.full_like(tier, s), tier)
        return FinalMsg(n_acc.contiguous(), drawn.contiguous(), tier.contiguous())

    def commit(self, dm: DraftMsg, final: FinalMsg) -> None:
        self.st.commit(self.ops, dm.tok, final)


# ------------------------------------------------------------------------------------------- heads
class LogitsHead:
    """Materialised logits: lm_head GEMM (rocBLAS) -> asd_verify_accept on the [n,K,V] tensor."""

    def __init__(self, model, ops):
        self.m, self.ops = model, ops
        self._logits = None

    def score(self, hid, tok, lp_d, u, inv_t):
        self._logits = self.m.lm_head(hid) * self.m.logit_scale          # [n, K+1, V], row K = bonus row
        K = tok.shape[1]
        return self.ops.verify_accept(self._logits[:, :K].contiguous(), tok, lp_d, u, inv_temperature=inv_t)

    def score_and_stop(self, hid, tok, lp_d, u, inv_t, **stop):
        """The tier step's ONE hot-path launch: verify + accept + the predictor / Bayes / DP epilogue inside the same
        kernel (ops.verify_stop -> asd_verify_accept_fused_ex)."""
        self._logits = self.m.lm_head(hid) * self.m.logit_scale
        K = tok.shape[1]
        return self.ops.verify_stop(self._logits[:, :K].contiguous(), tok, lp_d, u, inv_t, **stop)

    def draw_rows(self, sel: torch.Tensor, j: torch.Tensor) -> torch.Tensor:
        """Target logits row j[i] of local sequence sel[i] -> [m, V]."""
        return self._logits[sel, j].contiguous()


class FusedHead:
    """N2: asd_lm_head_verify on the hidden states (no [n,K,V] logits); the one row per sequence a draw needs
    is a skinny GEMM."""

    def __init__(self, model, ops):
        self.m, self.ops = model, ops
        self._hid = None

    def score(self, hid, tok, lp_d, u, inv_t):
        self._hid = hid
        K = tok.shape[1]
        return self.ops.lm_head_verify(hid[:, :K].contiguous(), self.m.lm_head.weight, tok, lp_d, u,
                                       inv_temperature=float(np.float32(inv_t * self.m.logit_scale)))

    def draw_rows(self, sel, j):
        return (self.m.lm_head(self._hid[sel, j]) * self.m.logit_scale).contiguous()


class ShardedHead:
    """The tier's lm_head split along the vocabulary over `group` (72B over 2 / 4 ranks): every rank holds rows
    [v0, v1) of the matrix (`model.lm_head.weight` is the SHARD), reduces them to (m2, s, g) triples without forming
    logits (asd_lm_head_partial), and the group all-gathers [n,K,3] floats.  Draw rows are all-gathered shard pieces."""

    def __init__(self, model, ops, vocab: int, group=None):
        from ..distributed import VocabShardedVerifier
        self.m, self.ops, self.group = model, ops, group
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.v0, self.v1 = shard_bounds(vocab, self.world, self.rank)
        self.vocab = vocab
        self.ver = VocabShardedVerifier(vocab, ops=ops, group=group)
        self._hid = None
        self.bytes_exchanged = 0

    def score(self, hid, tok, lp_d, u, inv_t):
        self._hid = hid
        K = tok.shape[1]
        self.ver.inv_temperature = float(np.float32(inv_t * self.m.logit_scale))
        self.bytes_exchanged += tok.shape[0] * K * 12 * (self.world - 1)
        return self.ver.verify_hidden(hid[:, :K].contiguous(), self.m.lm_head.weight, tok, lp_d, u)

    def draw_rows(self, sel, j):
        piece = (self.m.lm_head(self._hid[sel, j]) * self.m.logit_scale).contiguous()     # [m, v1-v0]
        width = max(shard_bounds(self.vocab, self.world, r)[1] - shard_bounds(self.vocab, self.world, r)[0]
                    for r in range(self.world))
        pad = torch.zeros((piece.shape[0], width), dtype=piece.dtype, device=piece.device)
        pad[:, :piece.shape[1]] = piece
        parts = all_gather_any(pad, self.group)
        self.bytes_exchanged += pad.numel() * pad.element_size() * (self.world - 1)
        cols = [p[:, :shard_bounds(self.vocab, self.world, r)[1] - shard_bounds(self.vocab, self.world, r)[0]]
                for r, p in enumerate(parts)]
        return torch.cat(cols, 1).contiguous()

    def draw_rows_owned(self, j_all, counts):
        """The batch-sharded target's form: sequence b belongs to ONE rank (rank q owns `counts[q]` consecutive sequences), so
        the shard pieces are exchanged with one all-to-all -- rank q receives [counts[q], V/N] from every rank, (N - 1) / N of
        ITS rows -- instead of all-gathering every sequence's row to every rank.  Returns the rank's own rows [counts[me], V]."""
        from ..distributed import all_to_all_rows
        Bt = int(j_all.shape[0])
        sel = torch.arange(Bt, device=j_all.device)
        piece = (self.m.lm_head(self._hid[sel, j_all]) * self.m.logit_scale).contiguous()  # [B, v1-v0]: my columns of EVERY row
        widths = [shard_bounds(self.vocab, self.world, r)[1] - shard_bounds(self.vocab, self.world, r)[0] for r in range(self.world)]
        width = max(widths)
        if piece.shape[1] != width:
            pad = torch.zeros((Bt, width), dtype=piece.dtype, device=piece.device)
            pad[:, :piece.shape[1]] = piece
            piece = pad
        if self.world == 1:
            return piece[:, :widths[0]].contiguous()
        got = all_to_all_rows(piece, counts, self.group)                                   # [N * mine, width]
        mine = int(counts[self.rank])
        self.bytes_exchanged += (Bt - mine) * width * piece.element_size()                 # what this rank SENDS
        got = got.view(self.world, mine, width)
        return torch.cat([got[r, :, :widths[r]] for r in range(self.world)], 1).contiguous()


# ------------------------------------------------------------------------------------------- tiers s >= 1
class VerifyRole:
    def __init__(self, model, stage_idx: int, cfg: HierarchyConfig, ops, prompt_ids: torch.Tensor, max_new_tokens: int,
                 predictor, head=None, feat: Optional[torch.Tensor] = None, keep_inputs: bool = False):
        self.m, self.s, self.cfg, self.ops = model, stage_idx, cfg, ops
        self.st = _SeqState(prompt_ids, max_new_tokens, cfg.draft_len)
        dev = prompt_ids.device
        B = self.st.B
        self.gen = _dev_gen(dev, cfg.seed + 7919 * stage_idx)
        self.inv_t = float(np.float32(1.0 / cfg.temperature))
        self.L = len(cfg.stage_costs)
        self.last = stage_idx == self.L - 1
        self.costs = torch.tensor(list(cfg.stage_costs), dtype=torch.float64, device=dev)
        self.pred = ops.pack_predictor(predictor, dev)
        self.feat = prompt_features(prompt_ids) if feat is None else feat
        self.head = head if head is not None else LogitsHead(model, ops)
        self.keep_inputs = keep_inputs
        model.reset()
        model.alloc_ragged(B, self.st.kv_slots)
        model.forward_ragged(prompt_ids[:, :self.st.P - 1], torch.zeros((B,), dtype=torch.int64, device=dev), self.st.P)
        self.kv_len = torch.full((B,), self.st.P - 1, dtype=torch.int64, device=dev)   # KV valid for positions < kv_len
        self._fed: Optional[torch.Tensor] = None            # sequences fed this step (their KV advanced)
        self._pending = None
        self.fed_tokens = 0                                 # model positions computed (cost accounting)
        self.fwd_calls = 0                                  # model passes (loop roofline: each streams the tier's weights once)
        self.events: Optional[list] = None                  # set to [] to bracket the model passes with HIP events

    def _empty(self) -> Verdict:
        z = torch.zeros_like(self.st.seq_len)
        return Verdict(z, z.clone(), z.clone())

    @torch.no_grad()
    def verify(self, dm: DraftMsg, esc: Optional[EscMsg]) -> Tuple[Verdict, EscMsg]:
        st, cfg, K = self.st, self.cfg, self.cfg.draft_len
        B, dev = st.B, st.tokens.device
        self._pending = None
        self._fed = None
        if esc is None:                                     # tier 1 sees every block that was not committed at stage 0
            act = dm.stop0 == 0
            p_prev = dm.p0[:, None]
        else:
            act = esc.escalate == 1
            p_prev = esc.p_prev
        idx = act.nonzero()[:, 0]                           # host sync: the size of this tier's sub-batch
        n = idx.numel()
        out_esc = EscMsg(torch.zeros((B,), dtype=torch.uint8, device=dev),
                         torch.ones((B, self.s + 1), dtype=torch.float64, device=dev))
        if n == 0:
            return self._empty(), out_esc
        L = st.seq_len.to(torch.int64)[idx]
        kv = self.kv_len[idx]
        lag = L - 1 - kv                                    # committed tokens this tier has not seen yet
        every_step = self.s == 1 and cfg.min_verify_stage >= 1           # tier 1 then sees every block: lag == 0
        T = K + 1 if every_step else int(lag.max().item()) + K + 1
        pos = kv[:, None] + torch.arange(T, device=dev)
        committed = st.tokens[idx].gather(1, pos.clamp(max=st.cap - 1)).to(torch.int64)
        tok_i = dm.tok[idx]
        drafted = tok_i.gather(1, (pos - L[:, None]).clamp(0, K - 1)).to(torch.int64)
        ids = torch.where(pos < L[:, None], committed, torch.where(pos < (L + K)[:, None], drafted, torch.zeros_like(drafted)))
        hid = _timed(self, lambda: self.m.forward_ragged(ids, kv, st.window(), return_hidden=True, rows=None if n == B else idx))
        self.fed_tokens += n * T
        self.fwd_calls += 1
        sel = lag[:, None] + torch.arange(K + 1, device=dev)
        hid = hid.gather(1, sel[:, :, None].expand(-1, -1, hid.shape[-1]))              # [n, K+1, D]
        u = torch.rand((B, K), generator=self.gen, device=dev)[idx].contiguous()
        self._r = torch.rand((B,), generator=self.gen, device=dev)
        lp_d = dm.lp_d[idx].contiguous()
        tok_i = tok_i.contiguous()
        # stop rule: p_hist columns < s from the tiers below, column s from this tier's log-probs, priors 1.0 above
        ph = torch.ones((n, self.L), dtype=torch.float64, device=dev)
        ph[:, :self.s] = p_prev[idx, :self.s]
        stop_args = dict(pred=self.pred, feat=self.feat[idx].contiguous(), p_hist=ph, stage_idx=self.s, costs=self.costs,
                         lam=cfg.lambda_value, risk_adjustment=cfg.risk_adjustment, n_obs=cfg.n_obs, alpha=cfg.risk_alpha,
                         beta=cfg.risk_beta, stats_col=cfg.stats_col)
        if hasattr(self.head, "score_and_stop"):            # materialised logits: verify + stop rule in ONE launch
            (lp_t, accept, n_acc, bits), (score, k_star, ph) = self.head.score_and_stop(hid, tok_i, lp_d, u, self.inv_t, **stop_args)
        else:                                               # hidden-state heads (fused GEMM, vocabulary shards): two launches
            lp_t, accept, n_acc, bits = self.head.score(hid, tok_i, lp_d, u, self.inv_t)
            score, k_star, ph = self.ops.predictor_stop(self.pred, lp_t.contiguous(), stop_args["feat"], ph, self.s,
                                                        self.costs, cfg.lambda_value, cfg.risk_adjustment, cfg.n_obs,
                                                        cfg.risk_alpha, cfg.risk_beta, cfg.stats_col)
        stop = torch.ones((n,), dtype=torch.bool, device=dev) if self.last else (k_star <= self.s)
        z = torch.zeros((B,), dtype=torch.int32, device=dev)
        v = Verdict(z.index_put((idx,), torch.ones((n,), dtype=torch.int32, device=dev)),
                    z.index_put((idx,), stop.to(torch.int32)), z.index_put((idx,), n_acc.to(torch.int32)),
                    idx=idx, accept=accept, k_star=k_star, score=score, p_hist=ph)
        if self.keep_inputs:
            v.inputs = dict(tok=tok_i.clone(), lp_d=lp_d.clone(), u=u.clone(), lp_t=lp_t.clone(), n_acc=n_acc.clone(),
                            hidden=hid[:, :K].clone(), feat=self.feat[idx].clone())
            if isinstance(self.head, LogitsHead):
                v.inputs["logits"] = self.head._logits[:, :K].clone()
        out_esc.escalate[idx] = (~stop).to(torch.uint8)
        out_esc.p_prev[idx] = ph[:, :self.s + 1]
        self._fed = idx
        self._pending = (idx, stop, n_acc)
        return v, out_esc

    def rows_expected(self, v: Verdict) -> torch.Tensor:
        """b indices (ascending) whose draft row the draft rank will send: stopped here with a rejection."""
        return ((v.stop == 1) & (v.n_acc < self.cfg.draft_len)).nonzero()[:, 0]

    @torch.no_grad()
    def draw(self, b_rows: torch.Tensor, d_rows: torch.Tensor, d_thr: torch.Tensor,
             only: Optional[Tuple[int, int]] = None) -> torch.Tensor:
        """The token every sequence that STOPPED at this tier commits after its accepted prefix -> drawn [B] i32
        (0 elsewhere).  b_rows / d_rows / d_thr: DraftRole.rows_for of this tier's verdict.  only = (b0, b1): draw
        for sequences b0 <= b < b1 only (the rank's own slice under replicated drafts; the target rows are still
        gathered for every stopping sequence, the gather being a collective of the tier's ranks)."""
        B, dev, K = self.st.B, self.st.tokens.device, self.cfg.draft_len
        drawn = torch.zeros((B,), dtype=torch.int32, device=dev)
        if self._pending is None:
            return drawn
        idx, stop, n_acc = self._pending
        sel = stop.nonzero()[:, 0]                          # local positions of the stopping sequences
        m = sel.numel()
        if m == 0:
            return drawn
        b_sel = idx[sel]
        j = n_acc.to(torch.int64)[sel]
        t_rows = self.head.draw_rows(sel, j)                # [m, V]: row n_acc (the bonus row when n_acc == K)
        if only is not None:
            mine = (b_sel >= only[0]) & (b_sel < only[1])
            b_sel, j, t_rows = b_sel[mine], j[mine], t_rows[mine].contiguous()
            m = b_sel.numel()
            if m == 0:
                return drawn
        d_full = torch.zeros_like(t_rows)
        thr = torch.full((m,), float("-inf"), dtype=torch.float32, device=dev)
        if b_rows.numel():
            if d_rows.dtype != t_rows.dtype:
                # the nucleus threshold was taken on the draft row AS STORED: a cast could move tokens across it
                raise ValueError(f"draft rows are {d_rows.dtype}, this tier's logits {t_rows.dtype}: the tiers of a hierarchy "
                                 "must produce logits of one storage dtype")
            where = torch.searchsorted(b_sel, b_rows)       # b_sel ascending (idx and sel are)
            d_full[where] = d_rows
            thr[where] = d_thr
        all_acc = (j >= K).to(torch.int32)                  # K = 1 view: 0 -> residual of the two rows, 1 -> bonus draw
        tokd = self.ops.residual_sample(t_rows[:, None, :], d_full[:, None, :], all_acc.contiguous(), self._r[b_sel].contiguous(),
                                        t_rows, self.inv_t, d_threshold=thr[:, None].contiguous())
        drawn[b_sel] = tokd.to(torch.int32)
        return drawn

    def commit(self, dm: DraftMsg, final: FinalMsg) -> None:
        cand = None
        if self._fed is not None:                           # fed positions L .. L+K-1 hold the drafted tokens' KV
            cand = self.st.seq_len.to(torch.int64)[self._fed] + final.n_acc.to(torch.int64)[self._fed]
        self.st.commit(self.ops, dm.tok, final)
        if cand is not None:                                # ... of which the accepted prefix stays valid (the commit may
            new_len = self.st.seq_len.to(torch.int64)[self._fed]            # have been cut short at the buffer's end)
            self.kv_len[self._fed] = torch.minimum(cand, new_len - 1)
        self._fed = self._pending = None                    # a step in which this tier is not called feeds nothing


def calibrate_lambda(ops, p_hist: torch.Tensor, costs: torch.Tensor, stage_idx: int, target_stop_rate: float,
                     lo: float = 0.05, hi: float = 500.0, grid: int = 256, rounds: int = 4) -> Tuple[float, float]:
    """A lambda controller over the device-side sweep (N4; the reference tunes lambda offline with
    algorithms/optimizer.py:47-205): given the p_hist rows [n, L] of blocks judged up to `stage_idx` (columns above
    it at their prior 1.0), each round is ONE asd_lambda_sweep launch that evaluates the DP rule for `grid`
    log-spaced lambdas; the share of blocks with k* <= stage_idx falls as lambda grows, and the next round refines
    the bracket in which it crosses `target_stop_rate` (scores of similar blocks differ in the 4th digit, so one
    coarse grid would only ever see "all stop" / "none stops").  Returns (lambda, share at that lambda)."""
    best_lam, best_share = lo, 1.0
    for _ in range(rounds):
        lams = torch.logspace(np.log10(lo), np.log10(hi), grid, dtype=torch.float64, device=p_hist.device)
        k = ops.lambda_sweep(p_hist.contiguous(), costs, lams)                    # [G, n]
        share = (k <= stage_idx).to(torch.float64).mean(dim=1)
        i = int((share - target_stop_rate).abs().argmin().item())
        best_lam, best_share = float(lams[i].item()), float(share[i].item())
        if abs(best_share - target_stop_rate) <= 0.5 / max(1, p_hist.shape[0]):
            break
        above = (share >= target_stop_rate).nonzero()                              # lambdas that still stop enough blocks
        j = int(above.max().item()) if above.numel() else 0
        if j + 1 >= grid:
            break
        lo, hi = float(lams[j].item()), float(lams[j + 1].item())
    return best_lam, best_share


class StopRateController:
    """calibrate_lambda as a RUNNING controller (the token-level counterpart of DynamicLambdaController /
    src/serving/dynamic_cost_optimizer.py:425-487, which nudges lambda from rolling statistics): it keeps the p_hist rows
    of the blocks tier `stage_idx` judged during the last `window` steps and, every `every` steps, re-solves -- ONE
    asd_lambda_sweep launch per refinement round over that window -- for the lambda whose stop share at that tier is
    `target_stop_rate`; the roles' shared HierarchyConfig.lambda_value is updated in place.  A one-shot calibration on a
    probe step drifts as soon as the population of blocks moves (round 2: target 0.66, live 0.48)."""

    def __init__(self, ops, cfg: HierarchyConfig, costs: torch.Tensor, stage_idx: int = 1, target_stop_rate: float = 0.66,
                 window: int = 4, every: int = 1, lo: float = 0.05, hi: float = 500.0):
        self.ops, self.cfg, self.costs, self.s = ops, cfg, costs, stage_idx
        self.target, self.window, self.every = float(target_stop_rate), int(window), int(every)
        self.lo, self.hi = lo, hi
        self.rows: List[torch.Tensor] = []
        self.steps = 0
        self.history: List[Tuple[float, float]] = []        # (lambda, share of the window that stops at it)

    def observe(self, p_hist: Optional[torch.Tensor]) -> None:
        """p_hist [n, L] of the blocks the tier judged this step (columns above it at their prior)."""
        if p_hist is not None and p_hist.shape[0]:
            self.rows.append(p_hist.detach().clone())
            del self.rows[:-self.window]

    def end_step(self) -> float:
        self.steps += 1
        if self.rows and self.steps % self.every == 0:
            lam, share = calibrate_lambda(self.ops, torch.cat(self.rows, 0), self.costs, self.s, self.target, self.lo, self.hi)
            self.cfg.lambda_value = lam
            self.history.append((lam, share))
        return self.cfg.lambda_value


# ------------------------------------------------------------------------------------------- drivers
@dataclass
class HierarchyTrace:
    tokens: torch.Tensor                    # [B, P + max_new] int32 (prompt + committed)
    seq_len: torch.Tensor
    steps: int = 0
    verified_tokens: int = 0                # tokens appended, all sequences, all steps
    tier_counts: List[int] = field(default_factory=list)     # sequences-steps whose verdict came from tier s
    tier_calls: List[int] = field(default_factory=list)      # sequence-steps each tier verified
    fed_tokens: List[int] = field(default_factory=list)      # model positions each verify tier computed
    records: List[dict] = field(default_factory=list)        # per step (keep_inputs)
    bytes_sent: Dict[str, int] = field(default_factory=dict)
    messages_sent: Dict[str, int] = field(default_factory=dict)     # backend calls per message name (one per message and destination)
    rows_shipped: int = 0
    tier_forwards: List[int] = field(default_factory=list)   # model forward passes per tier (tier 0: K per step)
    tier_forward_positions: List[int] = field(default_factory=list)   # sequence-positions those passes computed
    lambda_history: List[float] = field(default_factory=list)         # lambda in force at every step (StopRateController)

    @property
    def stop_rate(self) -> List[float]:
        tot = max(1, sum(self.tier_counts))
        return [c / tot for c in self.tier_counts]


def _check_status(ops) -> None:
    """Once per step, where the drivers synchronise anyway: a kernel that lost a hand-off poisoned its outputs AND raised its
    workspace's sticky status word (HipOps.check_status raises kernels.LostHandoffError); the oracle-backed test ops have none."""
    chk = getattr(ops, "check_status", None)
    if chk is not None:
        chk()


def _account(tr: HierarchyTrace, L: int, final: FinalMsg, verdicts) -> None:
    t = final.tier.cpu().numpy()
    for s in range(L):
        tr.tier_counts[s] += int((t == s).sum())
    for s, v, _ in verdicts:
        tr.tier_calls[s] += int(v.active.sum().item())


@torch.no_grad()
def generate_hierarchical(draft: DraftRole, tiers: Sequence[VerifyRole], max_steps: Optional[int] = None,
                          keep_inputs: bool = False, controller: Optional["StopRateController"] = None) -> HierarchyTrace:
    """All roles in ONE process (one GPU holds every tier): the reference configuration of the multi-rank run.
    controller: a StopRateController over the roles' shared config -- lambda then follows the running stop share."""
    L = draft.L
    assert len(tiers) == L - 1
    tr = HierarchyTrace(draft.st.tokens, draft.st.seq_len, tier_counts=[0] * L, tier_calls=[0] * L)
    cap = draft.st.cap
    limit = max_steps if max_steps is not None else cap + 4
    while tr.steps < limit:
        dm = draft.propose()
        verdicts, esc = [], None
        rec = dict(draft=dm, tiers={})
        for t in tiers:
            v, esc_out = t.verify(dm, esc)
            if controller is not None and t.s == controller.s:
                controller.observe(v.p_hist)
            if int(v.active.sum().item()) > 0:
                b_rows, d_rows, d_thr = draft.rows_for(v)
                tr.rows_shipped += int(b_rows.numel())
                drawn = t.draw(b_rows, d_rows, d_thr)
                verdicts.append((t.s, v, drawn))
                rec["tiers"][t.s] = (v, drawn)
            esc = esc_out
            if not bool(esc.escalate.any().item()):
                break
        final = draft.assemble(dm, verdicts)
        rec["final"] = final
        before = int(draft.st.seq_len.sum().item())
        draft.commit(dm, final)
        for t in tiers:
            t.commit(dm, final)
        tr.verified_tokens += int(draft.st.seq_len.sum().item()) - before
        _check_status(draft.ops)
        _account(tr, L, final, verdicts)
        if keep_inputs:
            tr.records.append(rec)
        tr.steps += 1
        tr.lambda_history.append(float(draft.cfg.lambda_value))
        if controller is not None:
            controller.end_step()
        if int(draft.st.seq_len.min().item()) >= cap:
            break
    tr.fed_tokens = [t.fed_tokens for t in tiers]
    tr.tier_forwards = [draft.fwd_calls] + [t.fwd_calls for t in tiers]
    tr.tier_forward_positions = [draft.fwd_positions] + [t.fed_tokens for t in tiers]
    return tr


# ---- roles on different ranks -----------------------------------------------------------------------------
@dataclass
class Placement:
    """Which rank runs which role.  `tiers[s-1]` lists the ranks of verify tier s: one rank, or several when the
    tier's lm_head is vocab-sharded over them (the first is the tier's leader)."""
    draft: int
    tiers: List[List[int]]

    def leader(self, s: int) -> int:
        return self.tiers[s - 1][0]

    def ranks_of(self, s: int) -> List[int]:
        return self.tiers[s - 1]

    @staticmethod
    def for_world(world: int, n_tiers: int = 3) -> "Placement":
        """BASELINE configs[3] / the reference's configs/qwen3_models.yaml:5-53 (7B [0], 32B [1], 72B TP over the
        rest): 1 rank holds everything; 2 ranks = {7B + 32B | 72B}; >= 4 ranks = 7B, 32B, 72B sharded over the
        remaining ranks (4 of them at most, the reference's tensor_parallel_size)."""
        if n_tiers != 3:
            raise ValueError("placements are defined for the 7B / 32B / 72B hierarchy")
        if world == 1:
            return Placement(0, [[0], [0]])
        if world == 2:
            return Placement(0, [[0], [1]])
        if world == 3:
            return Placement(0, [[1], [2]])
        return Placement(0, [[1], list(range(2, min(world, 6)))])


class Wire:
    """Point-to-point movement of the fixed-shape messages between roles; roles that share a rank hand the
    tensors over directly.  ONE backend call per (message, destination): the tensors of a message are packed into one
    byte buffer (each segment padded to 8 bytes) -- over RCCL every send / recv is a kernel launch of its own, and on
    first use between two ranks also the lazy construction of their point-to-point communicator, which `warm_up`
    therefore forces before the first timed step.  Counts the bytes and the messages that really crossed a link.

    loopback=True (tests, one GPU): a message between two roles of the SAME rank also goes through the backend --
    one grouped isend + irecv of the rank to itself (ncclSend / ncclRecv inside one ncclGroup on RCCL) -- instead of the
    direct hand-over, so that the device-tensor transport of the `nccl` backend executes on a box with a single GPU.
    The received copy replaces the original; the committed stream must not change (tests/test_gpu_hierarchy.py)."""

    def __init__(self, rank: int, device, group=None, loopback: bool = False):
        self.rank, self.device, self.group = rank, device, group
        self.staged = dist.is_initialized() and host_staged(group) and torch.device(device).type == "cuda"
        self.loopback = bool(loopback) and dist.is_initialized()
        self.local: Dict[Tuple[str, int, int], List[torch.Tensor]] = {}
        self.bytes: Dict[str, int] = {}          # name -> bytes handed to the backend (padding included)
        self.messages: Dict[str, int] = {}       # name -> backend calls (one per message and destination)

    # ---- packing: [tensor, ...] <-> one uint8 buffer
    @staticmethod
    def _seg(numel: int, esz: int) -> int:
        return (numel * esz + 7) // 8 * 8

    def _pack(self, tensors: Sequence[torch.Tensor]) -> torch.Tensor:
        total = sum(self._seg(t.numel(), t.element_size()) for t in tensors)
        dev = tensors[0].device if tensors else self.device
        buf = torch.zeros((total,), dtype=torch.uint8, device=dev)
        o = 0
        for t in tensors:
            n = t.numel() * t.element_size()
            if n:
                buf[o:o + n] = t.contiguous().reshape(-1).view(torch.uint8)
            o += self._seg(t.numel(), t.element_size())
        return buf

    def _unpack(self, buf: torch.Tensor, like: Sequence[Tuple[Tuple[int, ...], torch.dtype]]) -> List[torch.Tensor]:
        out, o = [], 0
        for shape, dtype in like:
            numel = int(np.prod(shape)) if len(shape) else 1
            esz = torch.empty((), dtype=dtype).element_size()
            n = numel * esz
            t = buf[o:o + n].clone().view(dtype).reshape(shape) if n else torch.empty(shape, dtype=dtype, device=buf.device)
            out.append(t.to(self.device))
            o += self._seg(numel, esz)
        return out

    def _like_of(self, tensors: Sequence[torch.Tensor]):
        return [(tuple(t.shape), t.dtype) for t in tensors]

    def _count(self, name: str, nbytes: int) -> None:
        self.bytes[name] = self.bytes.get(name, 0) + nbytes
        self.messages[name] = self.messages.get(name, 0) + 1

    def send(self, name: str, tensors: Sequence[torch.Tensor], src: int, dsts: Sequence[int]) -> None:
        if self.rank != src:
            return
        buf = None
        for d in dsts:
            if d == src:
                self.local[(name, src, d)] = list(tensors)
                continue
            if buf is None:
                buf = self._pack(tensors)
                if buf.is_cuda and self.staged:
                    buf = buf.cpu()
            if buf.numel():
                dist.send(buf, dst=d, group=self.group)
                self._count(name, buf.numel())

    def _through_backend(self, name: str, tensors: Sequence[torch.Tensor]) -> List[torch.Tensor]:
        src = self._pack(tensors)
        if not src.numel():
            return list(tensors)
        if src.is_cuda and self.staged:
            src = src.cpu()
        if host_staged(self.group):             # gloo has no pair of a rank with itself: the host staging alone
            dst = src.clone()
        else:
            dst = torch.empty_like(src)
            me = dist.get_rank()                # P2POp peers are global ranks
            for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, src, me, self.group),
                                             dist.P2POp(dist.irecv, dst, me, self.group)]):
                w.wait()
        self._count(name + " (loopback)", src.numel())
        return self._unpack(dst, self._like_of(tensors))

    def recv(self, name: str, like: Sequence[Tuple[Tuple[int, ...], torch.dtype]], src: int) -> List[torch.Tensor]:
        if self.rank == src:
            got = self.local.pop((name, src, src))
            if self.loopback:
                got = self._through_backend(name, got)
            return got
        total = sum(self._seg(int(np.prod(shape)) if len(shape) else 1, torch.empty((), dtype=dtype).element_size())
                    for shape, dtype in like)
        buf = torch.empty((total,), dtype=torch.uint8, device="cpu" if self.staged else self.device)
        if total:
            dist.recv(buf, src=src, group=self.group)
        return self._unpack(buf, like)

    def warm_up(self, pairs: Sequence[Tuple[int, int]]) -> None:
        """One 8-byte message over every (src, dst) pair the loop will use, in the given order (the same on every rank), so
        that the backend's per-pair state -- RCCL builds a point-to-point communicator on first use -- exists before the
        first timed step.  Not counted in `bytes` / `messages`."""
        if not dist.is_initialized():
            return
        for a, b in pairs:
            if a == b:
                continue
            t = torch.zeros((8,), dtype=torch.uint8, device="cpu" if (self.staged or torch.device(self.device).type == "cpu") else self.device)
            if self.rank == a:
                dist.send(t, dst=b, group=self.group)
            elif self.rank == b:
                dist.recv(t, src=a, group=self.group)


def placement_pairs(placement: "Placement", L: int) -> List[Tuple[int, int]]:
    """Every (src, dst) rank pair run_hierarchical_rank sends over, in a fixed order."""
    D = placement.draft
    verify_ranks = sorted({r for t in placement.tiers for r in t})
    everyone = sorted({D} | set(verify_ranks))
    pairs: List[Tuple[int, int]] = []

    def add(a, b):
        if a != b and (a, b) not in pairs:
            pairs.append((a, b))
    for r in verify_ranks:
        add(D, r)                                            # draft, rows
    for s in range(1, L):
        add(placement.leader(s), D)                          # verdict, drawn
        if s > 1:
            for r in placement.ranks_of(s):
                add(placement.leader(s - 1), r)              # escalate
    for r in everyone:
        add(D, r)                                            # final
    return pairs


@torch.no_grad()
def run_hierarchical_rank(rank: int, placement: Placement, draft: Optional[DraftRole], tiers: Dict[int, VerifyRole],
                          B: int, K: int, L: int, V: int, logits_dtype: torch.dtype, cap: int, device,
                          max_steps: Optional[int] = None, group=None, keep_inputs: bool = False,
                          loopback: bool = False, controller: Optional["StopRateController"] = None) -> HierarchyTrace:
    """One rank of the multi-rank loop.  `draft` is the DraftRole if this rank hosts tier 0, `tiers` maps stage index
    -> VerifyRole for the verify tiers this rank hosts (leader or vocab shard).  Every rank executes the same step
    sequence; only the messages listed in the module docstring cross ranks.  The committed stream is identical to
    generate_hierarchical's for the same seeds (same arithmetic, same uniforms).

    Per step and tier s:  leader(s-1) -> ranks(s): escalate (always, possibly all zero);  if the tier has work:
    leader(s) -> D: verdict;  D -> ranks(s): draft rows of the stop-with-rejection sequences;  leader(s) -> D: drawn.
    D learns whether tier s+1 runs from tier s's verdict (active and not stopped), tier s+1 from the escalate
    message; both are the same data, so every rank takes the same branch."""
    wire = Wire(rank, device, group, loopback=loopback)
    D = placement.draft
    verify_ranks = sorted({r for t in placement.tiers for r in t})
    everyone = sorted({D} | set(verify_ranks))
    if rank not in everyone:                  # a rank the placement gives no role (world > 6): nothing to run, nothing to wait for
        empty = torch.zeros((0,), dtype=torch.int32, device=device)
        return HierarchyTrace(empty, empty, tier_counts=[0] * L, tier_calls=[0] * L)
    state = draft.st if draft is not None else next(iter(tiers.values())).st
    tr = HierarchyTrace(state.tokens, state.seq_len, tier_counts=[0] * L, tier_calls=[0] * L)
    limit = max_steps if max_steps is not None else cap + 4
    i32, f32, f64, u8 = torch.int32, torch.float32, torch.float64, torch.uint8
    if state.steps == 0:                      # the first call on these roles: per-pair backend state before any timed step
        wire.warm_up(placement_pairs(placement, L))
    if controller is not None and len(placement.ranks_of(controller.s)) != 1:
        # lambda changes on the controller tier's leader only (nothing is broadcast): a tier whose verdict is computed on
        # several ranks (vocab shards) would then decide stop-or-escalate from different lambdas and desynchronise the wire
        raise ValueError(f"the lambda controller's tier ({controller.s}) is placed on {len(placement.ranks_of(controller.s))} ranks; "
                         "it must be a single-rank tier")
    while tr.steps < limit:
        # ---- tier 0 proposes; the block goes to every rank that hosts a verify tier
        dm = None
        if rank == D:
            dm = draft.propose()
            wire.send("draft", [dm.tok, dm.lp_d, dm.p0, dm.stop0], D, verify_ranks)
        if rank in verify_ranks:
            t_, l_, p_, s_ = wire.recv("draft", [((B, K), i32), ((B, K), f32), ((B,), f64), ((B,), u8)], D)
            dm = DraftMsg(t_, l_, p_, s_)
        verdicts = []
        rec = dict(draft=dm, tiers={})
        runs_d = bool((dm.stop0 == 0).any().item()) if rank == D else False      # D's view: does the next tier run?
        esc_out: Optional[EscMsg] = None                                         # what the tier below handed up
        for s in range(1, L):
            ranks_s = placement.ranks_of(s)
            lead = ranks_s[0]
            esc_in: Optional[EscMsg] = None
            if s > 1:
                prev = placement.leader(s - 1)
                if rank == prev:
                    wire.send("escalate", [esc_out.escalate, esc_out.p_prev], prev, ranks_s)
                if rank in ranks_s:
                    e_, p_ = wire.recv("escalate", [((B,), u8), ((B, s), f64)], prev)
                    esc_in = EscMsg(e_, p_)
            v = drawn = None
            runs_t = False
            if rank in ranks_s:
                v, esc_out = tiers[s].verify(dm, esc_in)
                runs_t = bool(v.active.any().item())
                # the controller lives where its tier does: lambda only enters that tier's stop-or-escalate decision
                # (tier 0 never stops a block with min_verify_stage = 1, the last tier always does), so nothing is broadcast
                if controller is not None and s == controller.s and rank == lead:
                    controller.observe(v.p_hist)
                if runs_t and rank == lead:
                    wire.send("verdict", [v.active, v.stop, v.n_acc], lead, [D])
            if rank == D and runs_d:
                a_, st_, n_ = wire.recv("verdict", [((B,), i32), ((B,), i32), ((B,), i32)], lead)
                vd = v if v is not None else Verdict(a_, st_, n_)
                b_rows, d_rows, d_thr = draft.rows_for(vd)
                tr.rows_shipped += int(b_rows.numel())
                wire.send("rows", [d_rows, d_thr], D, ranks_s)
            if rank in ranks_s and runs_t:
                b_exp = tiers[s].rows_expected(v)
                d_rows, d_thr = wire.recv("rows", [((b_exp.numel(), V), logits_dtype), ((b_exp.numel(),), f32)], D)
                drawn = tiers[s].draw(b_exp, d_rows, d_thr)
                if rank == lead:
                    wire.send("drawn", [drawn], lead, [D])
            if rank == D and runs_d:
                (dr_,) = wire.recv("drawn", [((B,), i32)], lead)
                verdicts.append((s, vd, dr_))
                rec["tiers"][s] = (vd, dr_)
                runs_d = bool(((vd.active == 1) & (vd.stop == 0)).any().item())
            elif rank in ranks_s and runs_t:
                verdicts.append((s, v, drawn))
                rec["tiers"][s] = (v, drawn)
        # ---- the final verdict reaches every rank
        if rank == D:
            final = draft.assemble(dm, verdicts)
            wire.send("final", [final.n_acc, final.drawn, final.tier], D, everyone)
        n_, d_, t_ = wire.recv("final", [((B,), i32), ((B,), i32), ((B,), i32)], D)
        final = FinalMsg(n_, d_, t_)
        rec["final"] = final
        before = int(state.seq_len.sum().item())
        if draft is not None:
            draft.commit(dm, final)
        for t in tiers.values():
            t.commit(dm, final)
        tr.verified_tokens += int(state.seq_len.sum().item()) - before
        _check_status(draft.ops if draft is not None else next(iter(tiers.values())).ops)
        _account(tr, L, final, verdicts)
        if keep_inputs:
            tr.records.append(rec)
        tr.steps += 1
        if controller is not None and controller.s in tiers and rank == placement.leader(controller.s):
            tr.lambda_history.append(float(tiers[controller.s].cfg.lambda_value))
            controller.end_step()
        if int(state.seq_len.min().item()) >= cap:
            break
    tr.fed_tokens = [tiers[s].fed_tokens if s in tiers else 0 for s in range(1, L)]
    tr.tier_forwards = [draft.fwd_calls if draft is not None else 0] + [tiers[s].fwd_calls if s in tiers else 0 for s in range(1, L)]
    tr.tier_forward_positions = [draft.fwd_positions if draft is not None else 0] + tr.fed_tokens
    tr.bytes_sent = dict(wire.bytes)
    tr.messages_sent = dict(wire.messages)
    return tr


# ---- construction helper shared by bench.py, tools/ and the tests --------------------------------------------
def build_rank_roles(rank: int, placement: Placement, shapes: Sequence, cfg: HierarchyConfig, prompt_ids: torch.Tensor,
                     max_new_tokens: int, predictor, ops=None, dtype: torch.dtype = torch.bfloat16,
                     heads: Sequence[str] = ("logits", "fused"), logit_scale: float = 1.0, seeds: Sequence[int] = (1, 2, 3),
                     keep_inputs: bool = False, weight_noise: Sequence[float] = (0.0, 0.0, 0.0), share_seed: Optional[int] = None,
                     hip_layers: Optional[bool] = None, pack_weights: bool = False, backend: Optional[str] = None
                     ) -> Tuple[Optional[DraftRole], Dict[int, VerifyRole]]:
    """The roles `rank` hosts under `placement`: tier 0 + verify tiers, models built on prompt_ids.device.
    shapes: one synthetic_lm.LMShape per tier.  heads[s-1]: "logits" (lm_head GEMM + asd_verify_accept), "fused"
    (asd_lm_head_verify from hidden states); a tier placed on several ranks gets a ShardedHead over those ranks
    (every rank of the tier must call this function: it creates the tier's process group).
    share_seed: all tiers start from the same seed (+ per-tier lm_head noise `weight_noise`) -- small test models
    that agree often enough to accept tokens.
    hip_layers: run the models' passes through asd_decoder_forward (serving/hip_decoder.py) instead of torch modules.  None:
    wherever the stack supports the model (CUDA, bf16, head_dim 128 -- every Qwen2.5 shape); True: required (raises
    otherwise); False: torch modules.  `model.execution` on every role's model says which one it got.
    backend: the backend of the process group a vocab-sharded tier creates ("nccl" when the job's default group is a gloo
    control group and the data path is RCCL); None = the default group's."""
    from .synthetic_lm import SyntheticLM
    ops = ops if ops is not None else HipOps()
    dev = prompt_ids.device
    L = len(cfg.stage_costs)
    assert len(shapes) == L

    def make(i):
        m = SyntheticLM(shapes[i], dtype=dtype, device=dev, seed=share_seed if share_seed is not None else seeds[i],
                        logit_scale=logit_scale)
        sh = shapes[i]
        # what asd_decoder_forward's check_shape accepts (csrc/decoder.hip): auto mode falls back to the torch modules for
        # anything else instead of raising at the first pass; hip_layers=True keeps raising
        can = (dev.type == "cuda" and dtype == torch.bfloat16 and sh.head_dim == 128 and sh.hidden == sh.heads * sh.head_dim
               and sh.hidden % 64 == 0 and sh.intermediate % 64 == 0 and sh.hidden <= 8192 and sh.heads % sh.kv_heads == 0)
        if hip_layers or (hip_layers is None and can):
            # pack_weights: the projection matrices re-laid tile-major in place (HipDecoder) where every row count allows it
            packable = pack_weights and all(n % 256 == 0 for n in (shapes[i].hidden, shapes[i].hidden + 2 * shapes[i].kv_heads * shapes[i].head_dim,
                                                                  2 * shapes[i].intermediate))
            m.enable_hip_layers(pack_weights=packable)
        if weight_noise[i]:
            g = torch.Generator(device=dev).manual_seed(1000 + i)
            with torch.no_grad():
                w = m.lm_head.weight
                w.add_((torch.randn(w.shape, generator=g, device=dev) * weight_noise[i]).to(w.dtype))
        return m

    draft = DraftRole(make(0), cfg, ops, prompt_ids, max_new_tokens, predictor) if rank == placement.draft else None
    tiers: Dict[int, VerifyRole] = {}
    for s in range(1, L):
        ranks_s = placement.ranks_of(s)
        group = dist.new_group(ranks_s, backend=backend) if len(ranks_s) > 1 else None          # collective: every rank calls it
        if rank not in ranks_s:
            continue
        m = make(s)
        if len(ranks_s) > 1:
            head = ShardedHead(m, ops, shapes[s].vocab, group=group)
            m.lm_head.weight = torch.nn.Parameter(m.lm_head.weight[head.v0:head.v1].clone(), requires_grad=False)
        elif heads[s - 1] == "fused":
            head = FusedHead(m, ops)
        else:
            head = LogitsHead(m, ops)
        tiers[s] = VerifyRole(m, s, cfg, ops, prompt_ids, max_new_tokens, predictor, head=head, keep_inputs=keep_inputs)
    return draft, tiers


# ---- BASELINE configs[4]: replicated drafts + ONE sharded target over all ranks -------------------------------
class ShardedTargetRole:
    """The target tier of BASELINE configs[4] on one rank of N (the reference shards its 72B stage with
    `tensor_parallel_size: 4`, configs/qwen3_models.yaml:34-51, passed to vLLM at src/serving/real_model_pipeline.py:98-108).

    MI355X placement: 288 GB of HBM hold a whole 72B body (143 GB of bf16 weights) next to the 7B draft, so the BODY is
    replicated and the WORK is sharded along the batch -- every rank feeds only its own [B/N, K+1] rows through its replica
    (per-rank KV cache and per-rank model time fall with N; round 3 ran the whole batch through every replica) -- while the
    lm_head, whose [V, D] matrix is the one operand every row multiplies, stays sharded along the VOCABULARY over all ranks
    (ShardedHead): each rank streams V/N rows of it for ALL sequences and reduces them to (m2, s, g) triples without forming
    logits.  Per step the ranks exchange

        all-gather  tok [B/N,K] i32, lp_d [B/N,K] f32, p_0 [B/N] f64                  the drafts                 ~1.2 KB / rank
        all-gather  final hidden states [B/N, K+1, D] (storage dtype)                  body -> sharded head       2.36 MB / rank (B/N = 16, D = 8192)
        all-gather  (m2, s, g) [B,K,3] f32                                             inside ShardedHead.score   12 KB at B = 128
        all-to-all  the shard pieces [B/N, V/N] of ONE target row per OWN sequence      inside ShardedHead.draw_rows_owned  4.3 MB / rank (B = 128, N = 8;
                                                                                       round 4a all-gathered every row to every rank: 34 MB)
        all-reduce  MIN of the shortest sequence's length (8 bytes: every rank leaves the loop at the same step)

    and never a [B,K,V] tensor; each rank draws and commits only its own sequences.  Uniforms are drawn for the WHOLE batch from one seeded generator on every rank (as
    VerifyRole does), so the committed stream does not depend on N (tests/test_hierarchy.py: bit-equal to the one-rank run
    at world size 2 and 4)."""

    def __init__(self, model, cfg: HierarchyConfig, ops, prompt_local: torch.Tensor, max_new_tokens: int, predictor,
                 head: "ShardedHead", b0: int, batch_total: int, group=None, feat_local: Optional[torch.Tensor] = None):
        self.m, self.s, self.cfg, self.ops, self.head, self.group = model, 1, cfg, ops, head, group
        self.b0, self.Bt = int(b0), int(batch_total)
        self.st = _SeqState(prompt_local, max_new_tokens, cfg.draft_len)
        self.b1 = self.b0 + self.st.B
        dev = prompt_local.device
        # sequences per rank of the head's group, in rank order (the all-to-all of the draw rows needs every rank's count)
        n_ranks = head.world
        if n_ranks > 1:
            mine = torch.tensor([self.st.B], dtype=torch.int64, device=dev)
            self.counts = [int(c.item()) for c in all_gather_any(mine, group)]
        else:
            self.counts = [self.st.B]
        assert sum(self.counts) == self.Bt and sum(self.counts[:head.rank]) == self.b0, "the ranks' slices must tile the batch in rank order"
        self.gen = _dev_gen(dev, cfg.seed + 7919 * 1)            # VerifyRole's generator of stage 1: the same uniforms
        self.inv_t = float(np.float32(1.0 / cfg.temperature))
        self.L = len(cfg.stage_costs)
        assert self.L == 2, "replicated drafts + one target: two tiers"
        self.costs = torch.tensor(list(cfg.stage_costs), dtype=torch.float64, device=dev)
        self.pred = ops.pack_predictor(predictor, dev)
        self.feat = prompt_features(prompt_local) if feat_local is None else feat_local
        model.reset()
        model.alloc_ragged(self.st.B, self.st.kv_slots)
        model.forward_ragged(prompt_local[:, :self.st.P - 1], torch.zeros((self.st.B,), dtype=torch.int64, device=dev), self.st.P)
        self.kv_len = torch.full((self.st.B,), self.st.P - 1, dtype=torch.int64, device=dev)
        self.fed_tokens = 0                                       # model positions THIS rank computed
        self.fwd_calls = 0
        self.events: Optional[list] = None
        self.bytes_exchanged = 0                                  # hidden states + drafts + drawn (the head counts its own)
        self._r = None

    @torch.no_grad()
    def body(self, dm_local: DraftMsg) -> torch.Tensor:
        """The rank's own [B/N, K+1] rows through its replica of the body -> final hidden states [B/N, K+1, D]
        (row j scores drafted token j, row K is the bonus row).  The tier sees every block, so its cache lags by
        exactly the last committed token."""
        st, K, dev = self.st, self.cfg.draft_len, self.st.tokens.device
        L = st.seq_len.to(torch.int64)
        kv = self.kv_len
        pos = kv[:, None] + torch.arange(K + 1, device=dev)
        committed = st.tokens.gather(1, pos.clamp(max=st.cap - 1)).to(torch.int64)
        drafted = dm_local.tok.gather(1, (pos - L[:, None]).clamp(0, K - 1)).to(torch.int64)
        ids = torch.where(pos < L[:, None], committed, drafted)
        hid = _timed(self, lambda: self.m.forward_ragged(ids, kv, st.window(), return_hidden=True))
        self.fed_tokens += st.B * (K + 1)
        self.fwd_calls += 1
        lag = L - 1 - kv                                          # 0 by construction; kept general
        sel = lag[:, None] + torch.arange(K + 1, device=dev)
        return hid.gather(1, sel[:, :, None].expand(-1, -1, hid.shape[-1])).contiguous()

    @torch.no_grad()
    def score(self, hid_all: torch.Tensor, dm_all: DraftMsg) -> Verdict:
        """Verdict of the WHOLE batch from the gathered hidden states (vocab-sharded head: every rank computes the same
        [B] verdict from the all-gathered triples); stop-rule trace for the rank's own slice."""
        K, dev = self.cfg.draft_len, hid_all.device
        Bt = self.Bt
        u = torch.rand((Bt, K), generator=self.gen, device=dev)
        self._r = torch.rand((Bt,), generator=self.gen, device=dev)
        lp_t, accept, n_acc, bits = self.head.score(hid_all, dm_all.tok.contiguous(), dm_all.lp_d.contiguous(), u, self.inv_t)
        sl = slice(self.b0, self.b1)
        ph = torch.ones((self.st.B, self.L), dtype=torch.float64, device=dev)
        ph[:, 0] = dm_all.p0[sl]
        score, k_star, ph = self.ops.predictor_stop(self.pred, lp_t[sl].contiguous(), self.feat, ph, 1, self.costs, self.cfg.lambda_value,
                                                    self.cfg.risk_adjustment, self.cfg.n_obs, self.cfg.risk_alpha,
                                                    self.cfg.risk_beta, self.cfg.stats_col)
        one = torch.ones((Bt,), dtype=torch.int32, device=dev)
        self._n_acc = n_acc
        return Verdict(one, one.clone(), n_acc.to(torch.int32), accept=accept, k_star=k_star, score=score, p_hist=ph)

    @torch.no_grad()
    def draw(self, b_rows_local: torch.Tensor, d_rows: torch.Tensor, d_thr: torch.Tensor) -> torch.Tensor:
        """The token each of the rank's OWN sequences commits behind its accepted prefix -> [B/N] i32.  The target row of
        every sequence is assembled from the shard pieces of all ranks (a collective of the group)."""
        K, dev = self.cfg.draft_len, self.st.tokens.device
        Bt = self.Bt
        j_all = self._n_acc.to(torch.int64)
        sl = slice(self.b0, self.b1)
        t_rows, j = self.head.draw_rows_owned(j_all, self.counts), j_all[sl]       # [B/N, V]: the rank's own rows
        m = self.st.B
        d_full = torch.zeros_like(t_rows)
        thr = torch.full((m,), float("-inf"), dtype=torch.float32, device=dev)
        if b_rows_local.numel():
            if d_rows.dtype != t_rows.dtype:
                raise ValueError(f"draft rows are {d_rows.dtype}, the target's logits {t_rows.dtype}: the tiers of a hierarchy "
                                 "must produce logits of one storage dtype")
            d_full[b_rows_local] = d_rows
            thr[b_rows_local] = d_thr
        all_acc = (j >= K).to(torch.int32)
        tokd = self.ops.residual_sample(t_rows[:, None, :], d_full[:, None, :], all_acc.contiguous(), self._r[sl].contiguous(),
                                        t_rows, self.inv_t, d_threshold=thr[:, None].contiguous())
        return tokd.to(torch.int32)

    def commit(self, dm_local: DraftMsg, final_local: FinalMsg) -> None:
        cand = self.st.seq_len.to(torch.int64) + final_local.n_acc.to(torch.int64)
        self.st.commit(self.ops, dm_local.tok, final_local)
        new_len = self.st.seq_len.to(torch.int64)
        self.kv_len = torch.minimum(cand, new_len - 1)


@torch.no_grad()
def run_sharded_target_rank(rank: int, world: int, draft: DraftRole, target: ShardedTargetRole, device,
                            max_steps: int, group=None) -> HierarchyTrace:
    """One rank of BASELINE configs[4]: draft the rank's own slice, run ITS rows through the target body, all-gather the
    hidden states for the vocab-sharded head, draw and commit the slice (ShardedTargetRole has the exchange list).  Two
    tiers (L = 2): the target's verdict is final.  The trace's tokens / seq_len are the rank's own sequences; callers that
    compare streams gather them (tests)."""
    from ..distributed import all_gather_any
    L = draft.L
    assert L == 2
    b0, b1 = target.b0, target.b1
    tr = HierarchyTrace(target.st.tokens, target.st.seq_len, tier_counts=[0] * L, tier_calls=[0] * L)
    while tr.steps < max_steps:
        dm_l = draft.propose()
        hid_l = target.body(dm_l)
        tok = torch.cat(all_gather_any(dm_l.tok, group), 0)
        lp_d = torch.cat(all_gather_any(dm_l.lp_d, group), 0)
        p0 = torch.cat(all_gather_any(dm_l.p0, group), 0)
        hid = torch.cat(all_gather_any(hid_l, group), 0)
        target.bytes_exchanged += (world - 1) * (hid_l.numel() * hid_l.element_size() + dm_l.tok.numel() * 8 + dm_l.p0.numel() * 8)
        dm = DraftMsg(tok, lp_d, p0, torch.zeros((tok.shape[0],), dtype=torch.uint8, device=device))
        v = target.score(hid, dm)
        v_local = Verdict(v.active[b0:b1], v.stop[b0:b1], v.n_acc[b0:b1])
        bl, d_rows, d_thr = draft.rows_for(v_local)
        tr.rows_shipped += int(bl.numel())
        drawn_l = target.draw(bl, d_rows, d_thr)
        final_l = FinalMsg(v_local.n_acc.contiguous(), drawn_l.contiguous(), torch.ones_like(v_local.n_acc))
        before = int(target.st.seq_len.sum().item())
        draft.commit(dm_l, final_l)
        target.commit(dm_l, final_l)
        tr.verified_tokens += int(target.st.seq_len.sum().item()) - before
        _check_status(target.ops)
        tr.tier_counts[1] += b1 - b0
        tr.tier_calls[1] += b1 - b0
        tr.steps += 1
        # every rank must leave the loop at the same step: the shortest sequence of the WHOLE batch decides (8 bytes, all-reduce MIN)
        shortest = target.st.seq_len.min().to(torch.int64).reshape(1)
        if world > 1:
            if shortest.is_cuda and host_staged(group):
                h = shortest.cpu()
                dist.all_reduce(h, op=dist.ReduceOp.MIN, group=group)
                shortest = h
            else:
                dist.all_reduce(shortest, op=dist.ReduceOp.MIN, group=group)
        if int(shortest.item()) >= target.st.cap:
            break
    tr.fed_tokens = [target.fed_tokens]
    tr.tier_forwards = [draft.fwd_calls, target.fwd_calls]
    tr.tier_forward_positions = [draft.fwd_positions, target.fed_tokens]
    return tr
