"""Mirror of the reference's src/algorithms/__init__.py:1-8 re-exports."""
from .dp_solver import (AdaptiveStopping, OptimalStoppingTable, bayesian_adjustment,  # noqa: F401
                        bayesian_adjustment_batch, compute_expected_cost, optimal_stopping_rule,
                        optimal_stopping_rule_batch)
from .optimizer import (GridSearchOptimizer, LambdaOptimizer, OptimizationResult, StagePopulation,  # noqa: F401
                        find_optimal_lambda)
