"""Per-token log-probabilities and the 64-d quality-predictor feature vector.

    token_logprobs      src/training/generate_training_data.py:128-136  (A6)
                        the reference loops `softmax(score[0]) -> probs[token_id] -> log -> .item()`
                        once per generated token (3 launches + 1 device->host sync each); here all T
                        tokens of all sequences are one asd_verify_accept launch.
    extract_features    src/training/generate_training_data.py:148-205  (A7)
                        text features are host string work exactly as in the reference; the five
                        log-prob statistics [5:10] come from asd_logprob_stats (numpy f64 semantics).
"""
from __future__ import annotations

from typing import Dict, List, Sequence

import numpy as np

from ..backend import get_backend

FEATURE_DIM = 64
_QUESTION_WORDS = ("what", "why", "how", "when", "where", "which")


def token_logprobs(scores, token_ids) -> np.ndarray:
    """log softmax(scores)[token] for every position.

    scores: [T, V] (or [B, T, V]) array / tensor of raw scores (`outputs.scores` stacked);
    token_ids: [T] (or [B, T]).  Returns float32 log-probs of the same leading shape."""
    import torch

    t = scores if isinstance(scores, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(scores))
    squeeze = t.dim() == 2
    if squeeze:
        t = t.unsqueeze(0)
    B, T, V = t.shape
    tok = np.asarray(token_ids.cpu() if isinstance(token_ids, torch.Tensor) else token_ids, dtype=np.int32).reshape(B, T)
    out = np.empty((B, T), np.float32)
    be = get_backend()
    for s in range(0, T, 64):                              # ASD_MAX_DRAFT_LEN positions per sequence per launch
        e = min(T, s + 64)
        r = be.verify_accept(t[:, s:e].contiguous(), tok[:, s:e], np.zeros((B, e - s), np.float32),
                             np.ones((B, e - s), np.float32))
        out[:, s:e] = r["lp_t"]
    return out[0] if squeeze else out


def token_logprobs_from_hidden(hidden, lm_head_weight, token_ids, inv_temperature: float = 1.0):
    """token_logprobs without the scores: log softmax(hidden @ lm_head_weight.T)[token] straight from the
    final hidden states (asd_lm_head_verify), so [T, V] scores are never stored.

    hidden: [T, D] or [B, T, D] bf16 CUDA tensor; lm_head_weight: [V, D] bf16 CUDA tensor (nn.Linear layout);
    token_ids: [T] or [B, T].  Returns a float32 CUDA tensor of log-probs of the leading shape."""
    import torch

    from .. import kernels as K

    squeeze = hidden.dim() == 2
    h3 = hidden.unsqueeze(0) if squeeze else hidden
    B, T, _ = h3.shape
    tok = torch.as_tensor(token_ids, device=h3.device).to(torch.int32).reshape(B, T)
    out = torch.empty((B, T), dtype=torch.float32, device=h3.device)
    vers = {}                                              # one workspace per chunk length (at most two: 64 and the rest)
    for s in range(0, T, 64):                              # ASD_MAX_DRAFT_LEN positions per sequence per launch
        e = min(T, s + 64)
        zeros = torch.zeros((B, e - s), dtype=torch.float32, device=h3.device)
        ver = vers.get(e - s)
        if ver is None:
            ver = vers[e - s] = K.LmHeadVerifier(lm_head_weight, B, e - s)
        r = ver(h3[:, s:e].contiguous(), tok[:, s:e].contiguous(), zeros, zeros + 1.0, inv_temperature=inv_temperature)
        out[:, s:e] = r.lp_target
    return out[0] if squeeze else out


def _text_features(prompt: str, output: str, metadata: Dict, stage_id: int) -> List[float]:
    pw, ow = prompt.split(), output.split()
    f: List[float] = [len(pw), len(prompt), len(ow), len(output), len(ow) / max(len(pw), 1)]
    f.extend([0.0] * 5)                                    # [5:10] filled by the caller
    f.append(len(set(ow)) / max(len(ow), 1))               # vocabulary diversity
    onehot = [0.0] * 4
    onehot[stage_id] = 1.0
    f.extend(onehot)
    gen_time = metadata.get("generation_time", 1.0)
    f.append(metadata.get("completion_tokens", 0) / max(gen_time, 0.001))
    f.append(int("def " in prompt or "```" in prompt or "import " in prompt))
    f.append(int(any(c in prompt for c in "+=*/<>")))
    low = prompt.lower()
    f.append(sum(1 for w in _QUESTION_WORDS if w in low))
    f.extend([0.0] * (FEATURE_DIM - len(f)))
    return f[:FEATURE_DIM]


def extract_features_batch(prompts: Sequence[str], outputs: Sequence[str], metadatas: Sequence[Dict],
                           stage_ids: Sequence[int]) -> np.ndarray:
    """[N, 64] float64; one asd_logprob_stats launch for the whole batch."""
    n = len(prompts)
    feats = np.array([_text_features(p, o, m, s) for p, o, m, s in zip(prompts, outputs, metadatas, stage_ids)],
                     dtype=np.float64).reshape(n, FEATURE_DIM)
    lens = np.array([len(m.get("logprobs", [])) for m in metadatas], dtype=np.int32)
    kmax = int(lens.max()) if n else 0
    if kmax > 0:
        lp = np.zeros((n, kmax), np.float32)
        for i, m in enumerate(metadatas):
            lp[i, :lens[i]] = np.asarray(m.get("logprobs", []), dtype=np.float32)
        feats[:, 5:10] = get_backend().logprob_stats(lp, lens)
    return feats


def extract_features(prompt: str, output: str, metadata: Dict, stage_id: int) -> List[float]:
    """The reference's signature: one sample in, list of 64 floats out."""
    return [float(x) for x in extract_features_batch([prompt], [output], [metadata], [stage_id])[0]]
