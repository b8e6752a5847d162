"""Threshold-based adaptive decoder -- API of the reference's src/minimal_adaptive_decoder.py, with
the predictor forward (A8), the stop test (A11) and the thresholds (A10) computed by libasd_hip.so.

    DecodingResult              minimal_adaptive_decoder.py:20-27
    MinimalQualityPredictor     :30-68   (state_dict keys net.0.weight/bias, net.3.weight/bias)
    MinimalAdaptiveDecoder      :71-223  (decode, set_lambda, _estimate_difficulty, _compute_regret)
    train_minimal_predictor     :226-270

Deliberate differences from the file as shipped, all needed for it to run at all:
  * the tokenizer is not fetched by name (`Qwen/Qwen3-7B` does not exist and there is no network):
    pass any object with `.encode(prompt, return_tensors="pt")`, default = `SimpleTokenizer`;
  * `_load_models` returns one descriptor per configured stage (the reference returns [], so its
    stage loop never executes and `selected_stage` stays None);
  * the predictor is used in eval mode (the reference leaves Dropout(0.1) live at decode, which
    makes its own score stochastic); parity is defined against `.eval()`;
  * `decode_batch` scores B prompts with one MLP launch and one stop-test launch.
"""
from __future__ import annotations

import re
import time
import zlib
from dataclasses import dataclass
from pathlib import Path
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn as nn
import yaml

from .backend import get_backend
from .theory.optimal_stopping import OptimalStoppingTheory, TheoreticalParameters

_FEATURE_DIM = 64
_MAX_TOKENS_FOR_LENGTH = 512


@dataclass
class DecodingResult:
    text: str
    selected_stage: int
    quality_estimate: float
    inference_time: float
    theoretical_regret: float


class SimpleTokenizer:
    """Deterministic offline stand-in for the shared tokenizer: words and punctuation marks hashed
    into a Qwen-sized id space.  Only `.encode(prompt, return_tensors="pt")` is provided."""

    vocab_size = 151936
    _pieces = re.compile(r"\w+|[^\w\s]")

    def encode(self, prompt: str, return_tensors: Optional[str] = "pt"):
        ids = [zlib.crc32(p.encode("utf-8")) % self.vocab_size for p in self._pieces.findall(prompt)]
        if return_tensors == "pt":
            return torch.tensor([ids], dtype=torch.int64)
        return ids


class MinimalQualityPredictor(nn.Module):
    """Linear(input_dim, hidden) -> ReLU -> Dropout(0.1) -> Linear(hidden, 1) -> Sigmoid.

    In eval mode `forward` is ONE launch of asd_mlp_predict (any batch size; CPU tensors are
    uploaded).  In train mode it is the plain autograd module (training is outside the hot path)."""

    def __init__(self, input_dim: int = 64, hidden_dim: int = 32):
        super().__init__()
        self.input_dim, self.hidden_dim = input_dim, hidden_dim
        self.net = nn.Sequential(nn.Linear(input_dim, hidden_dim), nn.ReLU(), nn.Dropout(0.1),
                                 nn.Linear(hidden_dim, 1), nn.Sigmoid())
        self._packed = None          # (version key, device tensor) cache of the packed weights

    # -- weights as the kernels want them
    def _weight_key(self):
        return tuple((p._version, p.data_ptr()) for p in self.parameters())

    def weights_numpy(self):
        sd = self.state_dict()
        return (sd["net.0.weight"].detach().cpu().numpy(), sd["net.0.bias"].detach().cpu().numpy(),
                sd["net.3.weight"].detach().cpu().numpy(), sd["net.3.bias"].detach().cpu().numpy())

    def packed_weights(self, device=None) -> torch.Tensor:
        """Device buffer in asd_mlp_predict layout (W1T, b1, W2, b2); rebuilt when parameters change."""
        from . import kernels
        key = (self._weight_key(), str(device))
        if self._packed is None or self._packed[0] != key:
            self._packed = (key, kernels.pack_mlp_weights(*self.weights_numpy(), device=device))
        return self._packed[1]

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if self.training:
            return self.net(x)
        squeeze = x.dim() == 1
        x2 = x.reshape(-1, self.input_dim)
        if x2.is_cuda:
            from . import kernels
            out = kernels.mlp_predict(x2.contiguous().float(), self.packed_weights(x2.device), self.input_dim,
                                      self.hidden_dim)
        else:
            w1, b1, w2, b2 = self.weights_numpy()
            out = torch.from_numpy(get_backend().mlp_predict(x2.detach().numpy(), w1, b1, w2, b2))
        out = out.reshape(-1, 1)
        return out[0] if squeeze else out

    def extract_features(self, prompt: str, tokenizer) -> torch.Tensor:
        """[len/512, unique/len, words/100, 0 x 61]   (minimal_adaptive_decoder.py:51-68)."""
        tokens = tokenizer.encode(prompt, return_tensors="pt")
        return torch.from_numpy(features_from_token_ids(tokens[0].tolist(), prompt))


def features_from_token_ids(ids: Sequence[int], prompt: str) -> np.ndarray:
    """A9 on raw token ids: float32 [64]."""
    feats = np.zeros(_FEATURE_DIM, dtype=np.float32)
    length = min(len(ids), _MAX_TOKENS_FOR_LENGTH)
    distinct = len(set(ids))
    feats[0] = np.float32(length / _MAX_TOKENS_FOR_LENGTH)
    feats[1] = np.float32(distinct / length if length > 0 else 0)
    feats[2] = np.float32(len(prompt.split()) / 100)
    return feats


class MinimalAdaptiveDecoder:
    """Pick the first stage whose threshold the predicted quality clears (or the last stage)."""

    def __init__(self, config_path: str, tokenizer=None, predictor: Optional[MinimalQualityPredictor] = None):
        with open(config_path, "r") as f:
            self.config = yaml.safe_load(f)
        self.theory = self._init_theory()
        self.thresholds: Dict[int, float] = self.theory.derive_optimal_policy()
        self.models = self._load_models()
        self.tokenizer = tokenizer if tokenizer is not None else SimpleTokenizer()
        self.predictor = predictor if predictor is not None else MinimalQualityPredictor()
        self.predictor.eval()
        if predictor is None:
            self._load_predictor_weights()

    # -- construction helpers
    def _stages(self) -> List[dict]:
        return self.config["models"]["stages"]

    # quality / cost of a stage whose YAML entry lacks them (the reference's own configs/qwen3_models.yaml does,
    # and the reference then dies with KeyError): the theory's defaults by size (optimal_stopping.py:38-43)
    _SIZE_DEFAULTS = {"7b": (0.7, 1.0), "8b": (0.7, 1.0), "13b": (0.8, 1.6), "14b": (0.8, 2.0), "32b": (0.85, 4.5),
                      "34b": (0.85, 4.2), "70b": (0.9, 8.8), "72b": (0.9, 10.0)}

    def _stage_quality_cost(self, i: int, stage: dict):
        dq, dc = self._SIZE_DEFAULTS.get(str(stage.get("size_label", "")).lower(), (0.7 + 0.05 * i, float(2 ** i)))
        stage.setdefault("theoretical_quality", dq)
        stage.setdefault("relative_cost", dc)
        return stage["theoretical_quality"], stage["relative_cost"]

    def _init_theory(self) -> OptimalStoppingTheory:
        stages = self._stages()
        qc = [self._stage_quality_cost(i, s) for i, s in enumerate(stages)]
        return OptimalStoppingTheory(TheoreticalParameters(
            n_stages=len(stages),
            quality_bounds=[q for q, _ in qc],
            cost_ratios=[c for _, c in qc],
            lambda_param=1.0))

    def _load_models(self) -> List[dict]:
        """One descriptor per stage.  Generation itself is the callers' (Stage objects of the
        serving pipeline); this class only selects the stage, like the reference."""
        return [dict(index=i, model_path=s.get("model_path"), size_label=s.get("size_label"))
                for i, s in enumerate(self._stages())]

    def _load_predictor_weights(self):
        path = Path("checkpoints/minimal_predictor.pt")
        if path.exists():
            self.predictor.load_state_dict(torch.load(path, weights_only=True))

    # -- decisions
    def _theta_vector(self) -> np.ndarray:
        n = len(self.models)
        return np.array([float(self.thresholds.get(s, 0.0)) for s in range(n)], dtype=np.float64)

    def _scores(self, feats: np.ndarray) -> np.ndarray:
        w1, b1, w2, b2 = self.predictor.weights_numpy()
        return get_backend().mlp_predict(feats, w1, b1, w2, b2)

    def decode_batch(self, prompts: Sequence[str], max_tokens: int = 100) -> List[DecodingResult]:
        t0 = time.time()
        if not prompts:
            return []
        feats = np.stack([self.predictor.extract_features(p, self.tokenizer).numpy() for p in prompts])
        scores = self._scores(feats)                                   # one asd_mlp_predict launch
        stages = get_backend().threshold_stop(scores, self._theta_vector())   # one asd_threshold_stop launch
        results = []
        for prompt, q, s in zip(prompts, scores, stages):
            s = int(s)
            label = self._stages()[s]["size_label"]
            regret = self._compute_regret(s, self._estimate_difficulty(prompt))
            results.append(DecodingResult(text=f"[Generated with Qwen3-{label}]", selected_stage=s,
                                          quality_estimate=float(q), inference_time=time.time() - t0,
                                          theoretical_regret=regret))
        return results

    def decode(self, prompt: str, max_tokens: int = 100) -> DecodingResult:
        return self.decode_batch([prompt], max_tokens)[0]

    # -- analysis helpers (A12; host string heuristics exactly as the reference defines them)
    def _estimate_difficulty(self, prompt: str) -> float:
        words = prompt.split()
        long_words = sum(1 for w in words if len(w) > 8)
        questions = prompt.count("?") + prompt.count("how") + prompt.count("why")
        length_factor = min(len(words) / 50, 1.0)
        return min((long_words / 10 + questions / 5 + length_factor) / 3, 1.0)

    def _compute_regret(self, chosen_stage: int, true_difficulty: float) -> float:
        optimal = 3
        for stage, edge in enumerate((0.3, 0.5, 0.7)):
            if true_difficulty < edge:
                optimal = stage
                break
        stages = self._stages()
        cost_gap = stages[chosen_stage]["relative_cost"] - stages[optimal]["relative_cost"]
        quality_gap = stages[optimal]["theoretical_quality"] - stages[chosen_stage]["theoretical_quality"]
        return max(0, quality_gap + cost_gap / 10)

    def set_lambda(self, lambda_value: float):
        self.theory.params.lambda_param = lambda_value
        self.thresholds = self.theory.derive_optimal_policy()


def train_minimal_predictor(train_data: List[Dict], val_data: List[Dict], epochs: int = 50,
                            save_path: Optional[str] = "checkpoints/minimal_predictor.pt") -> MinimalQualityPredictor:
    """Adam(1e-3) + BCE over batches {'features', 'quality_labels'} (:226-270).  Training is plain
    PyTorch autograd; it is not part of the hot path."""
    predictor = MinimalQualityPredictor()
    optimizer = torch.optim.Adam(predictor.parameters(), lr=0.001)
    criterion = nn.BCELoss()
    for epoch in range(epochs):
        predictor.train()
        train_loss = 0.0
        for batch in train_data:
            optimizer.zero_grad()
            loss = criterion(predictor(batch["features"]), batch["quality_labels"])
            loss.backward()
            optimizer.step()
            train_loss += loss.item()
        predictor.net.eval()       # validation through the torch module as well: no GPU needed to train
        val_loss = 0.0
        with torch.no_grad():
            for batch in val_data:
                val_loss += criterion(predictor.net(batch["features"]), batch["quality_labels"]).item()
        if epoch % 10 == 0:
            print(f"Epoch {epoch}: Train Loss={train_loss:.4f}, Val Loss={val_loss:.4f}")
    predictor.eval()
    if save_path:
        Path(save_path).parent.mkdir(parents=True, exist_ok=True)
        torch.save(predictor.state_dict(), save_path)
    return predictor
