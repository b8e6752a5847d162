"""The pipeline's collaborators that the reference imports but does not ship:
`src/models/predictor.py` (`QualityPredictor`, `FeatureExtractor`) exist only as code blocks in
docs/guides/RESEARCH_PROTOCOL.md:315-409.  They are restated here from that specification (A14: doc
only, NOT a parity target -- there is no runnable reference to pin them to), with the MLP forward
served by asd_mlp_predict (generic 256 -> 128 -> 1 path).

    FeatureExtractor.extract(prompt, draft_output, draft_logprobs, stage_id) -> float[256]
        [0] entropy of the last 32 tokens' top-k log-probs  -mean_t sum_j exp(lp_tj) * lp_tj
        [1] prompt words / 2048      [2] output words / 512
        [3] mean over tokens of the max log-prob (-10 when there are none)     [4] stage_id / 4
        zero-padded to 256
    QualityPredictor(feature_dim=256): Linear(256,128) -> ReLU -> Dropout(0.1) -> Linear(128,1) -> Sigmoid
        state_dict keys mlp.0.weight/bias, mlp.3.weight/bias (what server.py:171-176 loads)
        predict(prompt=, draft_output=, draft_logprobs=, stage_id=, feature_extractor=) -> float
"""
from __future__ import annotations

from typing import Optional, Sequence

import numpy as np
import torch
import torch.nn as nn

from ..backend import get_backend

FEATURE_DIM = 256


class FeatureExtractor:
    def extract(self, prompt: str, draft_output: str, draft_logprobs, stage_id: int) -> np.ndarray:
        feats = np.zeros(FEATURE_DIM, dtype=np.float64)
        lps = [] if draft_logprobs is None else [np.atleast_1d(np.asarray(lp, dtype=np.float64)) for lp in draft_logprobs]
        if len(lps) > 0:
            feats[0] = -np.mean([np.sum(np.exp(lp) * lp) for lp in lps[-32:]])
            feats[3] = np.mean([np.max(lp) for lp in lps])
        else:
            feats[3] = -10.0
        feats[1] = len(prompt.split()) / 2048
        feats[2] = len(draft_output.split()) / 512
        feats[4] = stage_id / 4.0
        return feats


    @staticmethod
    def extract_device(row_max_lp: torch.Tensor, row_entropy: torch.Tensor, prompt_words, output_words, stage_id: int,
                       n_valid: Optional[torch.Tensor] = None) -> torch.Tensor:
        """The same 256-d vector for a BATCH, from what asd_verify_accept_stats left on the device: row_max_lp /
        row_entropy [B, T] per verified position (entropy over the whole vocabulary: the k = V form of the doc's top-k
        sum), prompt_words / output_words [B] (numbers).  No host round trip, no per-token Python.  n_valid [B]: positions
        that count (default all T); the entropy term averages the last <= 32 of them, as `extract` does."""
        Bv, T = row_max_lp.shape
        dev = row_max_lp.device
        nv = torch.full((Bv,), T, device=dev, dtype=torch.int64) if n_valid is None else n_valid.to(torch.int64)
        pos = torch.arange(T, device=dev)[None, :]
        valid = pos < nv[:, None]
        last = valid & (pos >= (nv - 32).clamp_min(0)[:, None])
        cnt = nv.clamp_min(1).to(torch.float64)
        feats = torch.zeros((Bv, FEATURE_DIM), dtype=torch.float64, device=dev)
        ent = torch.where(last, row_entropy.to(torch.float64), torch.zeros((), dtype=torch.float64, device=dev))
        feats[:, 0] = ent.sum(1) / last.sum(1).clamp_min(1)
        mx = torch.where(valid, row_max_lp.to(torch.float64), torch.zeros((), dtype=torch.float64, device=dev))
        feats[:, 3] = torch.where(nv > 0, mx.sum(1) / cnt, torch.full((Bv,), -10.0, dtype=torch.float64, device=dev))
        feats[:, 1] = torch.as_tensor(prompt_words, dtype=torch.float64, device=dev) / 2048
        feats[:, 2] = torch.as_tensor(output_words, dtype=torch.float64, device=dev) / 512
        feats[:, 4] = stage_id / 4.0
        return feats.to(torch.float32)


class QualityPredictor(nn.Module):
    def __init__(self, feature_dim: int = FEATURE_DIM, hidden_dim: int = 128):
        super().__init__()
        self.feature_dim, self.hidden_dim = feature_dim, hidden_dim
        self.feature_extractor = FeatureExtractor()
        self.mlp = nn.Sequential(nn.Linear(feature_dim, hidden_dim), nn.ReLU(), nn.Dropout(0.1),
                                 nn.Linear(hidden_dim, 1), nn.Sigmoid())
        self.eval()

    def _weights(self):
        sd = self.state_dict()
        return (sd["mlp.0.weight"].cpu().numpy(), sd["mlp.0.bias"].cpu().numpy(), sd["mlp.3.weight"].cpu().numpy(),
                sd["mlp.3.bias"].cpu().numpy())

    def forward(self, features: torch.Tensor) -> torch.Tensor:
        if self.training:
            return self.mlp(features)
        x = features.reshape(-1, self.feature_dim).detach().cpu().numpy().astype(np.float32)
        out = torch.from_numpy(get_backend().mlp_predict(x, *self._weights()))
        return out.reshape(*features.shape[:-1], 1)

    def predict_batch(self, prompts: Sequence[str], draft_outputs: Sequence[str], draft_logprobs: Sequence,
                      stage_ids: Sequence[int], feature_extractor: Optional[FeatureExtractor] = None) -> np.ndarray:
        """One asd_mlp_predict launch for a whole batch of requests."""
        fx = feature_extractor if feature_extractor is not None and hasattr(feature_extractor, "extract") else self.feature_extractor
        x = np.stack([fx.extract(p, o, lp, s) for p, o, lp, s in zip(prompts, draft_outputs, draft_logprobs, stage_ids)])
        return get_backend().mlp_predict(x.astype(np.float32), *self._weights())

    def predict(self, prompt: str, draft_output: str, draft_logprobs, stage_id: int,
                feature_extractor: Optional[FeatureExtractor] = None) -> float:
        return float(self.predict_batch([prompt], [draft_output], [draft_logprobs], [stage_id], feature_extractor)[0])
