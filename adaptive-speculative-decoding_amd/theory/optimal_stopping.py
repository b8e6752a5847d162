"""Stage thresholds of the optimal-stopping theory -- API of the reference's
src/theory/optimal_stopping.py.

    TheoreticalParameters            optimal_stopping.py:15-23
    OptimalStoppingTheory            optimal_stopping.py:26-128
        derive_optimal_policy        :45-82  -> asd_derive_thresholds (f64, -ffp-contract=off)
        _compute_improvement_probability :84-91
        compute_regret_bound         :93-112   (closed form, host)
        sample_complexity            :114-128  (closed form, host)
    RegretAnalyzer                   optimal_stopping.py:131-201  (host bookkeeping, A12)

The thresholds are an O(n) f64 recursion evaluated once per `set_lambda`; they are computed by
the library's host entry point so that the values the GPU stop test (asd_threshold_stop) compares
against are the bit-exact ones.  `thresholds_array()` returns them as the dense vector the
kernels take.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional

import numpy as np

from ..backend import get_backend

_DEFAULT_QUALITY = (0.7, 0.8, 0.85, 0.9)        # optimal_stopping.py:40
_DEFAULT_COST = (1.0, 2.0, 4.5, 10.0)           # optimal_stopping.py:43
_IMPROVE_SLOPE = 0.6                            # optimal_stopping.py:91


@dataclass
class TheoreticalParameters:
    n_stages: int = 4
    quality_bounds: List[float] = None
    cost_ratios: List[float] = None
    lambda_param: float = 1.0
    epsilon: float = 0.1
    delta: float = 0.05


class OptimalStoppingTheory:
    """State = stage index, actions = {continue, stop}, reward(stop at s) = q_s - lambda * c_s."""

    def __init__(self, params: TheoreticalParameters):
        self.params = params
        if params.quality_bounds is None:
            self.params.quality_bounds = list(_DEFAULT_QUALITY)
        if params.cost_ratios is None:
            self.params.cost_ratios = list(_DEFAULT_COST)

    def thresholds_array(self) -> np.ndarray:
        """theta[0..n) as float64 (theta[n-1] == 0), the layout asd_threshold_stop consumes."""
        n = self.params.n_stages
        return get_backend().derive_thresholds(self.params.quality_bounds[:n], self.params.cost_ratios[:n],
                                               self.params.lambda_param)

    def derive_optimal_policy(self) -> Dict[int, float]:
        """{stage: theta_stage}; stop at stage s when the predicted quality >= theta_s (:45-82)."""
        theta = self.thresholds_array()
        n = self.params.n_stages
        policy: Dict[int, float] = {}
        for s in range(n - 1, -1, -1):                 # same insertion order as the reference dict
            policy[s] = 0 if s == n - 1 else float(theta[s])
        return policy

    def _compute_improvement_probability(self, stage: int) -> float:
        return _IMPROVE_SLOPE * (1 - self.params.quality_bounds[stage])

    def compute_regret_bound(self, T: int) -> float:
        n = self.params.n_stages
        top = self.params.quality_bounds[-1]
        gaps = [top - q for q in self.params.quality_bounds[:-1]]
        constant = 2 * np.sqrt(n) * max(gaps) * np.sqrt(2)
        return constant * np.sqrt(T * np.log(T))

    def sample_complexity(self) -> int:
        eps, delta, n = self.params.epsilon, self.params.delta, self.params.n_stages
        return int(np.ceil(2 * np.log(2 * n / delta) / (eps ** 2)))


class RegretAnalyzer:
    """Per-decision regret against the difficulty oracle of :177-187."""

    def __init__(self, theory: OptimalStoppingTheory):
        self.theory = theory
        self.history: List[float] = []

    def _reward(self, stage: int) -> float:
        prm = self.theory.params
        return prm.quality_bounds[stage] - prm.lambda_param * prm.cost_ratios[stage]

    def _get_optimal_stage(self, difficulty: float) -> int:
        for stage, edge in enumerate((0.3, 0.5, 0.7)):
            if difficulty < edge:
                return stage
        return 3

    def compute_instantaneous_regret(self, chosen_stage: int, input_difficulty: float) -> float:
        regret = self._reward(self._get_optimal_stage(input_difficulty)) - self._reward(chosen_stage)
        self.history.append(regret)
        return regret

    def compute_cumulative_regret(self) -> float:
        return sum(self.history)

    def compute_average_regret(self) -> float:
        return float(np.mean(self.history)) if self.history else 0.0

    def theoretical_vs_empirical(self, T: Optional[int] = None) -> Dict[str, float]:
        if T is None:
            T = len(self.history)
        bound = self.theory.compute_regret_bound(T)
        empirical = self.compute_cumulative_regret()
        return {"theoretical_bound": bound, "empirical_regret": empirical,
                "ratio": empirical / bound if bound > 0 else 0, "gap": bound - empirical}
