"""Serving-side mirror of the reference's src/serving package (pipeline only; the HTTP shell,
cost optimiser and vLLM stages are out of scope)."""
from .cache import RequestCache  # noqa: F401
from .components import FeatureExtractor, QualityPredictor  # noqa: F401
from .pipeline import AdaptiveSpeculativePipeline, PipelineConfig, RequestResult  # noqa: F401
