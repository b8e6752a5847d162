"""Mirror of the reference's src/theory/__init__.py:8 re-exports."""
from .optimal_stopping import OptimalStoppingTheory, RegretAnalyzer, TheoreticalParameters  # noqa: F401
