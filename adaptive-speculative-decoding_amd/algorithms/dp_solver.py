"""Optimal-stopping decision arithmetic -- API of the reference's src/algorithms/dp_solver.py,
computed by the batched f64 kernels of libasd_hip.so (csrc/decision.hip).

Reference symbols mirrored (same names, argument meaning, return shapes, error behaviour):
    optimal_stopping_rule     dp_solver.py:12-71     -> asd_optimal_stopping
    compute_expected_cost     dp_solver.py:74-103    -> asd_expected_cost
    bayesian_adjustment       dp_solver.py:106-130   -> asd_bayes_adjust
    OptimalStoppingTable      dp_solver.py:133-210   (precompute = ONE batched launch per lambda)
    AdaptiveStopping          dp_solver.py:213-289   (host bookkeeping; no kernel)

The scalar functions keep the reference's list-in / tuple-out signatures and cross the C ABI
with B = 1; the `_batch` forms are what a serving loop should call (one launch for B requests
instead of B Python calls, cf. pipeline.py:234-256).
"""
from __future__ import annotations

import logging
from typing import Dict, List, Sequence, Tuple

import numpy as np

from ..backend import get_backend

logger = logging.getLogger(__name__)


# --------------------------------------------------------------------------- batched forms
def optimal_stopping_rule_batch(p, C: Sequence[float], lam: float, risk_adjustment: bool = False,
                                alpha: float = 1.0, beta: float = 1.0) -> Tuple[np.ndarray, np.ndarray]:
    """p: [B,L]; C: [L].  Returns (k_star int32 [B], J float64 [B,L+1])."""
    p = np.asarray(p, dtype=np.float64)
    if p.ndim != 2:
        raise ValueError("p must be [B, L]")
    if p.shape[1] != len(C):
        raise ValueError("p and C must have the same length")
    return get_backend().optimal_stopping(p, np.asarray(C, dtype=np.float64), float(lam), bool(risk_adjustment),
                                          float(alpha), float(beta))


def bayesian_adjustment_batch(p_hat, n_obs: int, alpha: float = 1.0, beta: float = 1.0) -> np.ndarray:
    return get_backend().bayes_adjust(np.asarray(p_hat, dtype=np.float64).reshape(-1), int(n_obs), float(alpha),
                                      float(beta))


# --------------------------------------------------------------------------- reference signatures
def optimal_stopping_rule(p: List[float], C: List[float], lam: float, risk_adjustment: bool = False,
                          alpha: float = 1.0, beta: float = 1.0) -> Tuple[int, List[float]]:
    """k*, J of backward induction over stop/continue costs (dp_solver.py:12-71).

    Raises ValueError when len(p) != len(C) (dp_solver.py:34-35)."""
    if len(p) != len(C):
        raise ValueError("p and C must have the same length")
    k, J = optimal_stopping_rule_batch([list(p)], C, lam, risk_adjustment, alpha, beta)
    return int(k[0]), [float(x) for x in J[0]]


def compute_expected_cost(p: List[float], C: List[float], lam: float, stopping_stage: int) -> float:
    """sum(C[:k+1]) + lam * (1 - prod(p[:k+1]))   (dp_solver.py:74-103)."""
    out = get_backend().expected_cost(np.asarray([list(p)], dtype=np.float64), np.asarray(C, dtype=np.float64),
                                      float(lam), np.asarray([stopping_stage], dtype=np.int32))
    return float(out[0])


def bayesian_adjustment(p_hat: float, n_obs: int, alpha: float = 1.0, beta: float = 1.0) -> float:
    """Posterior mean of a Beta(alpha, beta) prior after n_obs observations (dp_solver.py:106-130)."""
    return float(bayesian_adjustment_batch([p_hat], n_obs, alpha, beta)[0])


class OptimalStoppingTable:
    """Memo table of k* keyed by probabilities rounded to 2 decimals (dp_solver.py:133-210)."""

    _FALLBACK_COSTS = (1.0, 1.6, 4.2, 8.8)          # dp_solver.py:203

    def __init__(self, lambda_values: List[float], num_stages: int = 4):
        self.lambda_values = lambda_values
        self.num_stages = num_stages
        self.table: Dict[float, Dict[tuple, int]] = {}

    def precompute(self, cost_ratios: List[float], prob_grid: List[List[float]]):
        """One batched launch per lambda over the whole grid (the reference loops in Python)."""
        grid = np.asarray(prob_grid, dtype=np.float64)
        keys = [tuple(round(x, 2) for x in scenario) for scenario in prob_grid]
        for lam in self.lambda_values:
            ks, _ = optimal_stopping_rule_batch(grid, cost_ratios, lam)
            self.table[lam] = {key: int(k) for key, k in zip(keys, ks)}
        logger.info("Precomputed table for %d lambda values", len(self.lambda_values))

    def lookup(self, probabilities: List[float], lambda_value: float, fallback_to_dp: bool = True) -> int:
        nearest = min(self.lambda_values, key=lambda x: abs(x - lambda_value))
        key = tuple(round(x, 2) for x in probabilities)
        hit = self.table.get(nearest, {}).get(key)
        if hit is not None:
            return hit
        if fallback_to_dp:
            costs = list(self._FALLBACK_COSTS[:len(probabilities)])
            return optimal_stopping_rule(list(probabilities), costs, lambda_value)[0]
        return len(probabilities) - 1


class AdaptiveStopping:
    """Running per-stage reward averages with Hoeffding radii (dp_solver.py:213-289).  Host only."""

    def __init__(self, initial_lambda: float = 1.0, confidence_level: float = 0.1):
        self.lambda_value = initial_lambda
        self.confidence_level = confidence_level
        self.stage_counts = np.zeros(4)
        self.stage_rewards = np.zeros(4)
        self.total_steps = 0

    def update_statistics(self, chosen_stage: int, observed_quality: float, observed_latency: float):
        self.stage_counts[chosen_stage] += 1
        reward = observed_quality - self.lambda_value * (observed_latency / 1000.0)
        n = self.stage_counts[chosen_stage]
        self.stage_rewards[chosen_stage] = ((n - 1) * self.stage_rewards[chosen_stage] + reward) / n
        self.total_steps += 1

    def get_confidence_bounds(self, stage: int) -> Tuple[float, float]:
        n = self.stage_counts[stage]
        if n == 0:
            return -np.inf, np.inf
        radius = np.sqrt(-np.log(self.confidence_level / 2) / (2 * n))
        mean = self.stage_rewards[stage]
        return mean - radius, mean + radius

    def should_explore(self, stage: int) -> bool:
        if self.stage_counts[stage] < 10:
            return True
        uppers = [self.get_confidence_bounds(i)[1] for i in range(4)]
        return bool(uppers[stage] >= max(uppers) - 0.1)


__all__ = ["optimal_stopping_rule", "optimal_stopping_rule_batch", "compute_expected_cost", "bayesian_adjustment",
           "bayesian_adjustment_batch", "OptimalStoppingTable", "AdaptiveStopping"]
