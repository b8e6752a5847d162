"""adaptive-speculative-decoding_amd -- MI355X-native draft-verify / accept / optimal-stopping hot
path behind the API of sa2shun/adaptive-speculative-decoding.

Layout (mirrors the reference's `src/` for the modules on the path):
    algorithms/dp_solver.py        optimal_stopping_rule, bayesian_adjustment, ...   (A1-A4)
    theory/optimal_stopping.py     OptimalStoppingTheory, TheoreticalParameters, ... (A10, A12)
    minimal_adaptive_decoder.py    MinimalAdaptiveDecoder, MinimalQualityPredictor   (A8, A9, A11)
    serving/pipeline.py            AdaptiveSpeculativePipeline, PipelineConfig, ...  (A13)
    serving/speculative.py         token-level draft/verify/accept loop              (A5)
    training/logprobs.py           token_logprobs, extract_features                  (A6, A7)
    kernels.py / _binding.py       device-tensor front end / ctypes FFI of libasd_hip.so
    csrc/                          the HIP kernels and the C ABI (include/asd_hip.h)

Importing this package does not touch the GPU; the first computing call loads libasd_hip.so and
raises if it (or a GPU) is missing.  There is no CPU fallback.
"""
from ._binding import AsdError, LIB_PATH, load_library  # noqa: F401
from .backend import HipBackend, get_backend, set_backend  # noqa: F401

__version__ = "0.3.0"
