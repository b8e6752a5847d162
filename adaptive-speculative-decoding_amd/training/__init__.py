"""Mirror of the two hot-path functions of the reference's src/training/generate_training_data.py."""
from .logprobs import extract_features, extract_features_batch, token_logprobs  # noqa: F401
