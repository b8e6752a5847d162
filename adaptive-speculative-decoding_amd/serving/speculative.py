"""Token-level draft -> verify -> accept -> stop loop (the path BASELINE.json's north star names).

The reference has no token-level verification (its "speculative decoding" is a
model cascade), so this module DEFINES the step, on top of the two kernels that carry the
arithmetic:

  draft tier   proposes K tokens per sequence, keeping lp_d[b,k] = log q(tok[b,k])
  target tier  scores the K drafted positions in one forward: logits_t [B, K(+1), V]
  verify       asd_verify_accept: lp_t, accept[b,k] = log u <= lp_t - lp_d, n_acc[b], ballot word
  stop         asd_predictor_stop: log-prob statistics of the K target log-probs -> feature columns
               [5:10] -> quality predictor -> Bayes adjustment -> DP stop rule over the tiers.  The loops in
               THIS module have one verifying tier, so they only record the decision; the loop in which it
               acts -- sequences whose k* lies above the current tier are re-verified by the next tier -- is
               serving/hierarchy.py.
  commit       n_acc accepted tokens + one token from the target: asd_residual_sample draws it from the
               residual distribution max(0, p_t - p_d) at the first rejection, or from the target's own
               next-token distribution when all K pass.

`SpeculativeVerifier` is the device-resident state of one (draft tier, target tier) pair: work
space, packed predictor weights, stage costs, history of adjusted probabilities.  Everything it
does per step is two launches through the C ABI; no host synchronisation.

`speculative_generate` drives two `SyntheticLM`s through the loop.  Model execution is PLUMBING
(third-party in the reference); the proposal draw (asd_draft_sample), the verify step and the commit
draw (asd_residual_sample_ex) are kernels.  Lock-step commit of min_b(n_acc)+1 tokens so the batch
shares one KV length; `speculative_generate_ragged` commits per sequence.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Sequence

import torch

from . import synthetic_lm  # noqa: F401  (re-exported for callers)
from .. import kernels as K


@dataclass
class StepResult:
    verify: K.VerifyResult
    stop: Optional[K.StopResult]


class SpeculativeVerifier:
    def __init__(self, batch: int, draft_len: int, vocab: int, *, logits_dtype: torch.dtype = torch.bfloat16,
                 stage_costs: Sequence[float] = (1.0, 4.5, 10.0), lambda_value: float = 1.0,
                 risk_adjustment: bool = True, risk_alpha: float = 1.0, risk_beta: float = 1.0, n_obs: int = 100,
                 predictor=None, stats_col: int = 5, prefix_rule: bool = False, fused: bool = True, device=None):
        if draft_len > 64:
            raise ValueError("draft_len must be <= 64 (one ballot word per sequence)")
        self.B, self.Kd, self.V = batch, draft_len, vocab
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.ws = K.VerifyWorkspace(batch, draft_len, vocab, logits_dtype, self.device)
        self.L = len(stage_costs)
        self.costs = torch.tensor(list(stage_costs), dtype=torch.float64, device=self.device)
        self.lam, self.risk = float(lambda_value), bool(risk_adjustment)
        self.alpha, self.beta, self.n_obs = float(risk_alpha), float(risk_beta), int(n_obs)
        self.stats_col, self.prefix = stats_col, bool(prefix_rule)
        self.fused = bool(fused)          # one launch per step (asd_verify_accept_fused_ex; default) instead of two
        self.inv_temperature = 1.0        # sampling temperature of the tier pair, fused into the verify pass
        self.p_hist = torch.ones((batch, self.L), dtype=torch.float64, device=self.device)
        self.sampler = K.ResidualSampler(batch, vocab, logits_dtype, self.device)   # commit step (asd_residual_sample_ex)
        self.draft_sampler = K.DraftSampler(batch, vocab, logits_dtype, self.device)  # proposal step (asd_draft_sample)
        self.top_p = 1.0                  # nucleus of the draft tier (reference: 0.9, generate_training_data.py:110-119)
        self.in_dim = self.hidden = 0
        self.packed = None
        self._lm_head = None              # (key, kernels.LmHeadVerifier) of the last verify_hidden call
        if predictor is not None:
            self.set_predictor(predictor)

    def set_predictor(self, predictor) -> None:
        """predictor: a MinimalQualityPredictor (or anything with weights_numpy / input_dim / hidden_dim)."""
        self.in_dim, self.hidden = predictor.input_dim, predictor.hidden_dim
        self.packed = K.pack_mlp_weights(*predictor.weights_numpy(), device=self.device)

    def update_lambda(self, lam: float) -> None:
        self.lam = float(lam)

    def verify(self, logits: torch.Tensor, tok: torch.Tensor, lp_draft: torch.Tensor, u: torch.Tensor,
               out: Optional[K.VerifyResult] = None) -> K.VerifyResult:
        """`logits` are RAW target logits: 1/temperature is applied inside the kernel (self.inv_temperature)."""
        return K.verify_accept(logits, tok, lp_draft, u, self.ws, out, inv_temperature=self.inv_temperature)

    def verify_hidden(self, hidden: torch.Tensor, lm_head_weight: torch.Tensor, tok: torch.Tensor,
                      lp_draft: torch.Tensor, u: torch.Tensor, out: Optional[K.VerifyResult] = None,
                      logit_scale: float = 1.0) -> K.VerifyResult:
        """N2: verify from the target's final hidden states [B, K, D] and its lm_head matrix [V, D] (bf16);
        the [B, K, V] logits stay in MFMA accumulators (asd_lm_head_verify).  `logit_scale` multiplies the
        logits like SyntheticLM.logit_scale; it rides on the temperature constant."""
        key = (lm_head_weight.data_ptr(), tok.shape[0], tok.shape[1])
        if self._lm_head is None or self._lm_head[0] != key:
            self._lm_head = (key, K.LmHeadVerifier(lm_head_weight, tok.shape[0], tok.shape[1]))
        return self._lm_head[1](hidden, tok, lp_draft, u, out, inv_temperature=self.inv_temperature * logit_scale)

    def step_hidden(self, hidden: torch.Tensor, lm_head_weight: torch.Tensor, tok: torch.Tensor, lp_draft: torch.Tensor,
                    u: torch.Tensor, feat: Optional[torch.Tensor] = None, stage_idx: int = 0,
                    out: Optional[K.VerifyResult] = None, logit_scale: float = 1.0) -> StepResult:
        """`step` for a target tier that hands over hidden states instead of logits."""
        v = self.verify_hidden(hidden, lm_head_weight, tok, lp_draft, u, out, logit_scale)
        return StepResult(v, self._stop(v, tok, feat, stage_idx))

    def _stop(self, v: K.VerifyResult, tok: torch.Tensor, feat: Optional[torch.Tensor], stage_idx: int):
        if self.packed is None or feat is None:
            return None
        return K.predictor_stop(feat, self.packed, self.in_dim, self.hidden, stage_idx=stage_idx, L=self.L,
                                lp=v.lp_target, stats_col=self.stats_col, risk_adjustment=self.risk, n_obs=self.n_obs,
                                alpha=self.alpha, beta=self.beta, p_hist=self.p_hist[: tok.shape[0]], Cc=self.costs,
                                lam=self.lam, prefix_rule=self.prefix)

    def step(self, logits: torch.Tensor, tok: torch.Tensor, lp_draft: torch.Tensor, u: torch.Tensor,
             feat: Optional[torch.Tensor] = None, stage_idx: int = 0, out: Optional[K.VerifyResult] = None) -> StepResult:
        """One verify + stop decision for the whole batch: ONE launch (two with fused=False), nothing synchronises."""
        if self.fused and self.packed is not None and feat is not None:
            v, s = K.verify_accept_fused(logits, tok, lp_draft, u, self.ws, feat, self.packed, self.in_dim, self.hidden,
                                         stage_idx=stage_idx, L=self.L, stats_col=self.stats_col,
                                         risk_adjustment=self.risk, n_obs=self.n_obs, alpha=self.alpha, beta=self.beta,
                                         p_hist=self.p_hist[: tok.shape[0]], Cc=self.costs, lam=self.lam,
                                         prefix_rule=self.prefix, out=out, inv_temperature=self.inv_temperature)
            return StepResult(v, s)
        v = self.verify(logits, tok, lp_draft, u, out)
        return StepResult(v, self._stop(v, tok, feat, stage_idx))


# ------------------------------------------------------------------------------------ plumbing
def _propose(verifier: "SpeculativeVerifier", logits: torch.Tensor, gen: torch.Generator):
    """X1: logits [B,V] -> (tok [B] int64, log q(tok) [B] f32, nucleus threshold [B] f32) under the draft tier's
    softmax(logits / T) truncated to its top-p nucleus -- one asd_draft_sample call (the reference delegates this to
    HF generate(do_sample=True, temperature=0.7, top_p=0.9), generate_training_data.py:110-119)."""
    logits = logits.contiguous()
    r = torch.rand((logits.shape[0],), generator=gen, device=logits.device)
    d = verifier.draft_sampler(logits, r, verifier.inv_temperature, verifier.top_p)
    return d.tok.to(torch.int64), d.lp, d.thr


@dataclass
class GenerationTrace:
    tokens: torch.Tensor            # [B, T] committed tokens (prompt excluded)
    steps: int
    verified_tokens: int            # sum over steps and sequences of n_acc + 1
    accept_masks: List[torch.Tensor]
    step_inputs: List[dict]         # per step: logits/tok/lp_d/u as given to the verifier (for parity tests)
    stop_flags: List[Optional[torch.Tensor]]


@torch.no_grad()
def speculative_generate(draft, target, prompt_ids: torch.Tensor, max_new_tokens: int, verifier: SpeculativeVerifier,
                         *, temperature: float = 1.0, seed: int = 0, feat: Optional[torch.Tensor] = None,
                         keep_inputs: bool = False) -> GenerationTrace:
    """Drive two SyntheticLMs through draft/verify/accept/stop steps.  prompt_ids: [B, P] int64 on the GPU."""
    dev = prompt_ids.device
    gen = torch.Generator(device=dev).manual_seed(seed)
    B, Kd = prompt_ids.shape[0], verifier.Kd
    verifier.inv_temperature = 1.0 / temperature
    draft.reset()
    target.reset()
    seq = prompt_ids
    d_logits = draft(seq)[:, -1]                    # next-token logits after the prompt
    t_last = target(seq)[:, -1]                     # target's next-token logits for the first position
    out_tokens: List[torch.Tensor] = []
    masks, inputs, stops = [], [], []
    steps = verified = produced = 0
    while produced < max_new_tokens:
        base = seq.shape[1]
        toks, lps, dls, thrs, dl = [], [], [], [], d_logits
        for k in range(Kd):                         # draft K tokens autoregressively
            t, lp, thr = _propose(verifier, dl, gen)
            toks.append(t)
            lps.append(lp)
            thrs.append(thr)
            dls.append(dl)                          # kept for the residual distribution at a rejection
            if k + 1 < Kd:
                dl = draft(t[:, None])[:, -1]
        tok = torch.stack(toks, 1)
        lp_d = torch.stack(lps, 1).contiguous()
        d_thr = torch.stack(thrs, 1).contiguous()
        t_new = target(tok)                         # [B, K, V]: row k scores the token AFTER draft token k
        # logits that score draft position k: k = 0 -> t_last, k > 0 -> t_new[:, k-1]
        score = torch.cat([t_last[:, None], t_new[:, :-1]], dim=1).contiguous()   # raw logits: no scaling pass
        u = torch.rand((B, Kd), generator=gen, device=dev)
        res = verifier.step(score, tok.to(torch.int32).contiguous(), lp_d, u, feat)
        n_acc = res.verify.n_acc.to(torch.int64)
        m = int(n_acc.min().item()) + 1             # lock-step commit length (keeps one KV length per model)
        # the token every sequence emits after ITS accepted prefix: residual draw at the first rejection,
        # bonus draw from the target's next-token logits when all K passed -- one call, raw logits in
        r = torch.rand((B,), generator=gen, device=dev)
        drawn = verifier.sampler(score, torch.stack(dls, 1).to(score.dtype).contiguous(), res.verify.n_acc, r,
                                 bonus_logits=t_new[:, -1].contiguous(), inv_temperature=verifier.inv_temperature,
                                 d_threshold=d_thr)
        drawn = drawn.to(torch.int64)
        if m - 1 == Kd:                             # every sequence accepted all K: K drafts + the bonus token
            commit = torch.cat([tok, drawn[:, None]], 1)
        else:                                       # sequences that rejected at m-1 take their draw, the others keep the draft
            commit = tok[:, :m].clone()
            commit[:, m - 1] = torch.where(n_acc < m, drawn, commit[:, m - 1])
        masks.append(res.verify.accept.clone())
        stops.append(None if res.stop is None or res.stop.stop is None else res.stop.stop.clone())
        if keep_inputs:
            inputs.append(dict(logits=score.clone(), tok=tok.to(torch.int32).clone(), lp_d=lp_d.clone(), u=u.clone()))
        verified += int(n_acc.sum().item()) + B
        out_tokens.append(commit)
        produced += commit.shape[1]
        steps += 1
        # roll both KV caches back to the committed prefix and feed the committed tokens not yet cached
        seq = torch.cat([seq, commit], 1)
        # position of the last committed token differs per sequence (accepted draft vs resample): drop it too
        keep = base + commit.shape[1] - 1
        draft.truncate(min(draft.cached_len, keep))
        target.truncate(min(target.cached_len, keep))
        d_logits = draft(seq[:, draft.cached_len:])[:, -1]
        t_last = target(seq[:, target.cached_len:])[:, -1]
    return GenerationTrace(torch.cat(out_tokens, 1)[:, :max_new_tokens], steps, verified, masks, inputs, stops)


@dataclass
class RaggedTrace:
    tokens: torch.Tensor            # [B, P + max_new_tokens] int32: prompt + committed tokens, row b valid up to seq_len[b]
    seq_len: torch.Tensor           # [B] int32
    steps: int
    verified_tokens: int            # sum over steps and sequences of the tokens actually appended
    accept_masks: List[torch.Tensor]
    step_inputs: List[dict]
    commits: List[torch.Tensor]     # per step: n_commit [B]


@torch.no_grad()
def speculative_generate_ragged(draft, target, prompt_ids: torch.Tensor, max_new_tokens: int,
                                verifier: SpeculativeVerifier, *, temperature: float = 1.0, seed: int = 0,
                                feat: Optional[torch.Tensor] = None, keep_inputs: bool = False,
                                sync_every: int = 4, greedy_hidden: bool = False) -> RaggedTrace:
    """The loop with per-sequence lengths: every sequence commits ITS n_acc + 1 tokens per
    step (no lock-step minimum), both KV caches are per-sequence and rolling back after a rejection is the
    length update asd_commit_step does on the device.  The host reads nothing back inside a step; it looks
    at min(seq_len) every `sync_every` steps to decide whether to stop.

    Invariant at the top of a step, L = seq_len[b]: tokens[b, :L] are committed; the target's KV is valid
    for positions < L - 1 and the draft's for positions < L - 2 (at least), so the target is fed
    [t_{L-1}, d_0 .. d_{K-1}] at position L - 1 (row i scores d_i, row K is the bonus row) and the draft first
    re-feeds the last two committed tokens.

    greedy_hidden=True is greedy decoding with a target tier that never forms logits (N2 + N3): the draft
    proposes its arg-max tokens, the target hands over the K + 1 hidden rows, asd_lm_head_verify_ex accepts
    where the draft token IS the target's arg-max and reports every row's arg-max, and the token after the
    accepted prefix is argmax[b, n_acc[b]] (row K, whose draft slot holds -1, is the bonus row).  The output
    equals the target's own greedy continuation."""
    dev = prompt_ids.device
    gen = torch.Generator(device=dev).manual_seed(seed)
    B, P = prompt_ids.shape
    Kd = verifier.Kd
    if P < 2:
        raise ValueError("the ragged loop needs a prompt of at least two tokens")
    verifier.inv_temperature = 1.0 / temperature
    cap = P + max_new_tokens
    for m in (draft, target):
        m.reset()
        m.alloc_ragged(B, cap + Kd + 2)
    tokens = torch.zeros((B, cap), dtype=torch.int32, device=dev)
    tokens[:, :P] = prompt_ids.to(torch.int32)
    seq_len = torch.full((B,), P, dtype=torch.int32, device=dev)
    n_commit = torch.zeros((B,), dtype=torch.int32, device=dev)
    zero = torch.zeros((B,), dtype=torch.int64, device=dev)
    target.forward_ragged(prompt_ids[:, :P - 1], zero, P)            # prefill: everything but the last prompt token
    if P > 2:
        draft.forward_ragged(prompt_ids[:, :P - 2], zero, P)
    rows = torch.arange(B, device=dev)
    gv = g_arg = None
    masks, inputs, commits = [], [], []
    verified = torch.zeros((), dtype=torch.int64, device=dev)
    steps = 0
    while True:
        L = seq_len.to(torch.int64)
        window = min(P + steps * (Kd + 1) + Kd + 1, cap + Kd + 2)    # host-side bound on every position touched this step
        last2 = torch.stack([tokens[rows, L - 2], tokens[rows, L - 1]], 1).to(torch.int64)
        dl = draft.forward_ragged(last2, L - 2, window)[:, -1]
        toks, lps, dls, thrs = [], [], [], []
        for k in range(Kd):
            if greedy_hidden:
                t, lp, thr = dl.argmax(-1), torch.zeros((B,), device=dev), None
            else:
                t, lp, thr = _propose(verifier, dl, gen)
            toks.append(t)
            lps.append(lp)
            thrs.append(thr)
            dls.append(dl)
            if k + 1 < Kd:
                dl = draft.forward_ragged(t[:, None], L + k, window)[:, -1]
        tok = torch.stack(toks, 1)
        lp_d = torch.stack(lps, 1).contiguous()
        if greedy_hidden:
            hid = target.forward_ragged(torch.cat([last2[:, 1:], tok], 1), L - 1, window, return_hidden=True)   # [B, K+1, D]
            tok_pad = torch.cat([tok.to(torch.int32), torch.full((B, 1), -1, dtype=torch.int32, device=dev)], 1).contiguous()
            if gv is None:
                gv = K.LmHeadVerifier(target.lm_head.weight, B, Kd + 1)
                g_arg = torch.empty((B, Kd + 1), dtype=torch.int32, device=dev)
            vr = gv(hid, tok_pad, greedy=True, argmax_out=g_arg, inv_temperature=target.logit_scale)
            drawn = g_arg.gather(1, vr.n_acc.to(torch.int64)[:, None])[:, 0].contiguous()
            tok32 = tok_pad[:, :Kd].contiguous()
            K.commit_step(tok32, vr.n_acc, drawn, seq_len, tokens, n_commit, max_len=cap)
            verified += n_commit.sum()
            masks.append(vr.accept[:, :Kd].clone())
            commits.append(n_commit.clone())
            if keep_inputs:
                inputs.append(dict(tok=tok32.clone(), n_acc=vr.n_acc.clone(), drawn=drawn.clone(), argmax=g_arg.clone()))
            steps += 1
            if steps % sync_every == 0 and int(seq_len.min().item()) >= cap:
                break
            if steps > max_new_tokens + sync_every:
                raise RuntimeError("ragged loop did not terminate")
            continue
        t_out = target.forward_ragged(torch.cat([last2[:, 1:], tok], 1), L - 1, window)     # [B, K+1, V]
        score = t_out[:, :Kd].contiguous()
        u = torch.rand((B, Kd), generator=gen, device=dev)
        tok32 = tok.to(torch.int32).contiguous()
        res = verifier.step(score, tok32, lp_d, u, feat)
        r = torch.rand((B,), generator=gen, device=dev)
        drawn = verifier.sampler(score, torch.stack(dls, 1).to(score.dtype).contiguous(), res.verify.n_acc, r,
                                 bonus_logits=t_out[:, Kd].contiguous(), inv_temperature=verifier.inv_temperature,
                                 d_threshold=torch.stack(thrs, 1).contiguous())
        K.commit_step(tok32, res.verify.n_acc, drawn, seq_len, tokens, n_commit, max_len=cap)
        verified += n_commit.sum()
        masks.append(res.verify.accept.clone())
        commits.append(n_commit.clone())
        if keep_inputs:
            inputs.append(dict(logits=score.clone(), tok=tok32.clone(), lp_d=lp_d.clone(), u=u.clone(),
                               n_acc=res.verify.n_acc.clone(), drawn=drawn.clone()))
        steps += 1
        if steps % sync_every == 0 and int(seq_len.min().item()) >= cap:
            break
        if steps > max_new_tokens + sync_every:      # cannot happen: every step appends >= 1 token per unfinished row
            raise RuntimeError("ragged loop did not terminate")
    return RaggedTrace(tokens, seq_len, steps, int(verified.item()), masks, inputs, commits)
