"""Draft tier and target tier on DIFFERENT ranks (BASELINE configs[3]; SURVEY §8e "tiers on
different GPUs").  The verify kernel runs where the target logits are produced; per step the link
carries

    draft  -> target   tok [B,K] i32 + lp_d [B,K] f32                      (2 KB at B=32, K=8)
    target -> draft    accept [B,K] + n_acc [B]                            (~1.2 KB)
    draft  -> target   ONE draft-logits row per sequence, only when a position was rejected
                       (needed for the exact residual distribution max(0, p_t - p_d); B*V*2 bytes)
    target -> draft    the committed tokens [B, m]

and never the [B,K,V] target logits.  Both roles keep their own KV cache and roll it back to the
committed prefix.  Model execution and sampling are plain torch (plumbing, as in
serving/speculative.py); the arithmetic on the path goes through `ops` (distributed.HipOps on
the GPU box, the oracle in the gloo CPU tests).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List

import torch

from ..distributed import TierLink


def _sample(logits: torch.Tensor, temperature: float, gen: torch.Generator):
    lp = torch.log_softmax(logits.float() / temperature, dim=-1)
    tok = torch.multinomial(lp.exp(), 1, generator=gen)[:, 0]
    return tok, lp.gather(1, tok[:, None])[:, 0]


@dataclass
class TierTrace:
    tokens: torch.Tensor
    steps: int = 0
    verified_tokens: int = 0
    bytes_draft_to_target: int = 0
    bytes_target_to_draft: int = 0
    accept_masks: List[torch.Tensor] = field(default_factory=list)


def _refill(model, seq: torch.Tensor, base: int, committed: int) -> torch.Tensor:
    """Roll the KV cache back to the committed prefix minus its last token, feed the rest."""
    model.truncate(min(model.cached_len, base + committed - 1))
    return model(seq[:, model.cached_len:])[:, -1]


@torch.no_grad()
def run_draft_tier(draft, link: TierLink, prompt_ids: torch.Tensor, max_new_tokens: int, draft_len: int,
                   temperature: float = 1.0, seed: int = 0) -> TierTrace:
    dev = prompt_ids.device
    gen = torch.Generator(device=dev).manual_seed(seed)
    B, K = prompt_ids.shape[0], draft_len
    draft.reset()
    seq = prompt_ids
    d_logits = draft(seq)[:, -1]
    tr = TierTrace(tokens=prompt_ids[:, :0])
    out: List[torch.Tensor] = []
    produced = 0
    while produced < max_new_tokens:
        base = seq.shape[1]
        toks, lps, dls, dl = [], [], [], d_logits
        for k in range(K):
            t, lp = _sample(dl, temperature, gen)
            toks.append(t)
            lps.append(lp)
            dls.append(dl)
            if k + 1 < K:
                dl = draft(t[:, None])[:, -1]
        tok = torch.stack(toks, 1)
        lp_d = torch.stack(lps, 1).contiguous()
        link.send_draft(tok, lp_d)
        tr.bytes_draft_to_target += 8 * B * K
        accept, n_acc = link.recv_verdict(B, K, dev)
        tr.bytes_target_to_draft += 4 * (B * K + B)
        m = int(n_acc.min().item()) + 1
        if m - 1 < K:                                   # a rejection: the target needs q(.) at that position
            row = dls[m - 1].float().contiguous()
            link.send_to_target(row)
            tr.bytes_draft_to_target += row.numel() * 4
        width = m if m - 1 < K else K + 1
        commit = link.recv_from_target((B, width), torch.int64, dev)
        tr.bytes_target_to_draft += commit.numel() * 8
        tr.accept_masks.append(accept)
        tr.verified_tokens += int(n_acc.sum().item()) + B
        out.append(commit)
        produced += width
        tr.steps += 1
        seq = torch.cat([seq, commit], 1)
        d_logits = _refill(draft, seq, base, width)
    tr.tokens = torch.cat(out, 1)[:, :max_new_tokens]
    return tr


@torch.no_grad()
def run_target_tier(target, link: TierLink, prompt_ids: torch.Tensor, max_new_tokens: int, draft_len: int, ops,
                    temperature: float = 1.0, seed: int = 0) -> TierTrace:
    dev = prompt_ids.device
    gen = torch.Generator(device=dev).manual_seed(seed)
    B, K = prompt_ids.shape[0], draft_len
    V = target.shape.vocab
    target.reset()
    seq = prompt_ids
    t_last = target(seq)[:, -1]
    tr = TierTrace(tokens=prompt_ids[:, :0])
    out: List[torch.Tensor] = []
    produced = 0
    while produced < max_new_tokens:
        base = seq.shape[1]
        tok32, lp_d = link.recv_draft(B, K, dev)
        tok = tok32.to(torch.int64)
        t_new = target(tok)
        score = torch.cat([t_last[:, None], t_new[:, :-1]], dim=1).contiguous()   # raw logits: 1/T is applied in-kernel
        u = torch.rand((B, K), generator=gen, device=dev)
        _, accept, n_acc, _ = ops.verify_accept(score, tok32.contiguous(), lp_d.contiguous(), u,
                                                inv_temperature=1.0 / temperature)
        link.send_verdict(accept, n_acc)
        n_acc = n_acc.to(torch.int64)
        m = int(n_acc.min().item()) + 1
        if m - 1 == K:                                  # everything accepted everywhere: bonus token
            bonus, _ = _sample(t_new[:, -1], temperature, gen)
            commit = torch.cat([tok, bonus[:, None]], 1)
        else:
            d_row = link.recv_from_draft((B, V), torch.float32, dev)
            commit = tok[:, :m].clone()
            need = n_acc < m
            p_t = torch.softmax(score[:, m - 1].float() / temperature, -1)
            p_d = torch.softmax(d_row / temperature, -1)
            resid = (p_t - p_d).clamp_min(0)
            resid = torch.where(resid.sum(-1, keepdim=True) > 0, resid, p_t)
            rs = torch.multinomial(resid / resid.sum(-1, keepdim=True), 1, generator=gen)[:, 0]
            commit[:, m - 1] = torch.where(need, rs, commit[:, m - 1])
        link.send_to_draft(commit)
        tr.accept_masks.append(accept)
        tr.verified_tokens += int(n_acc.sum().item()) + B
        out.append(commit)
        produced += commit.shape[1]
        tr.steps += 1
        seq = torch.cat([seq, commit], 1)
        t_last = _refill(target, seq, base, commit.shape[1])
    tr.tokens = torch.cat(out, 1)[:, :max_new_tokens]
    return tr
