"""ctypes binding of libasd_hip.so (include/asd_hip.h).

This is the whole FFI: plain pointers and sizes, no torch types cross the boundary.  The library
is loaded from adaptive-speculative-decoding_amd/lib/ (built in-tree by build.py); if it is
missing or a symbol is absent the import of anything that computes FAILS LOUDLY -- there is no
CPU or PyTorch fallback anywhere in this package.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
# ASD_LIB_PATH: load another build of the SAME library (the host-sanitizer build of `make asan-host`); never a fallback
LIB_PATH = os.environ.get("ASD_LIB_PATH") or os.path.join(_HERE, "lib", "libasd_hip.so")

ASD_OK = 0
DTYPE_F32, DTYPE_BF16, DTYPE_F16 = 0, 1, 2
MAX_DRAFT_LEN = 64
MAX_STAGES = 16
MAX_SPLITS = 64
MAX_MLP_DIM = 1024
NUM_LP_STATS = 5
WS_LOST_HANDOFF = 0x1

_vp, _i, _i64, _d, _sz, _f = C.c_void_p, C.c_int, C.c_int64, C.c_double, C.c_size_t, C.c_float

# name -> (restype, argtypes); mirrors include/asd_hip.h declaration by declaration
SIGNATURES = {
    "asd_version": (_i, []),
    "asd_status_string": (C.c_char_p, [_i]),
    "asd_device_cu_count": (_i, [_i]),
    "asd_verify_accept_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "asd_workspace_init": (_i, [_vp, _sz, _vp]),
    "asd_workspace_status": (_i, [_vp, _vp, _vp]),
    "asd_verify_accept": (_i, [_vp, _i, _i64, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "asd_verify_accept_ex": (_i, [_vp, _i, _i64, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _sz, _vp, _vp]),
    "asd_verify_accept_stats": (_i, [_vp, _i, _i64, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _f, _vp]),
    "asd_verify_accept_fused": (_i, [_vp, _i, _i64, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _sz,
                                     _vp, _i64, _i, _vp, _i, _i, _i, _i64, _d, _d, _vp, _vp, _d, _i, _i, _i, _vp,
                                     _vp, _vp, _vp, _vp, _vp, _vp]),
    "asd_verify_accept_fused_ex": (_i, [_vp, _i, _i64, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _sz,
                                        _vp, _i64, _i, _vp, _i, _i, _i, _i64, _d, _d, _vp, _vp, _d, _i, _i, _i, _vp,
                                        _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "asd_residual_sample_workspace_bytes": (_sz, [_i, _i, _i]),
    "asd_residual_sample": (_i, [_vp, _i64, _vp, _i64, _vp, _i64, _i, _vp, _vp, _i, _i, _i, _f, _vp, _vp, _sz, _vp]),
    "asd_residual_sample_ex": (_i, [_vp, _i64, _vp, _i64, _vp, _i64, _i, _vp, _vp, _i, _i, _i, _f, _vp, _vp, _vp, _sz, _vp]),
    "asd_draft_sample_workspace_bytes": (_sz, [_i, _i, _i]),
    "asd_draft_sample": (_i, [_vp, _i64, _i, _vp, _i, _i, _f, _f, _vp, _vp, _vp, _vp, _sz, _vp]),
    "asd_lse_partial": (_i, [_vp, _i, _i64, _vp, _i, _i, _i, _i64, _f, _vp, _vp, _sz, _vp]),
    "asd_accept_from_partials": (_i, [_vp, _i, _vp, _vp, _i, _i, _f, _vp, _vp, _vp, _vp, _vp]),
    "asd_lm_head_verify_workspace_bytes": (_sz, [_i, _i, _i]),
    "asd_lm_head_verify": (_i, [_vp, _i64, _vp, _i64, _i, _i, _vp, _vp, _vp, _i, _i, _i, _f, _vp, _vp, _vp, _vp, _vp,
                                _sz, _vp]),
    "asd_lm_head_verify_ex": (_i, [_vp, _i64, _vp, _i64, _i, _i, _vp, _vp, _vp, _i, _i, _i, _f, _i, _vp, _vp, _vp, _vp, _vp,
                                   _vp, _sz, _vp]),
    "asd_lm_head_packed_bytes": (_sz, [_i, _i]),
    "asd_lm_head_pack_weights": (_i, [_vp, _i64, _i, _i, _i, _vp, _sz, _vp]),
    "asd_linear_workspace_bytes": (_sz, [_i, _i, _i]),
    "asd_linear": (_i, [_vp, _i64, _vp, _i64, _vp, _i, _i, _i, _i, _vp, _i64, _vp, _sz, _vp]),
    "asd_linear_ex": (_i, [_vp, _i64, _vp, _i64, _vp, _vp, _i64, _i, _i, _i, _i, _vp, _i64, _vp, _sz, _vp]),
    "asd_linear_partial": (_i, [_vp, _i64, _vp, _i64, _vp, _vp, _i64, _i, _i, _i, _i, _vp, _i64, _vp, _sz, _vp, _vp]),
    "asd_linear_slices": (_i, [_i, _i, _i]),
    "asd_rmsnorm": (_i, [_vp, _i64, _vp, _f, _i, _i, _i, _vp, _i64, _vp]),
    "asd_rope_kv_store": (_i, [_vp, _i64, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp]),
    "asd_attn_ragged": (_i, [_vp, _i64, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _i64, _vp]),
    "asd_silu_mul": (_i, [_vp, _i64, _i, _i, _i, _vp, _i64, _vp]),
    "asd_decoder_scratch_bytes": (_sz, [_vp, _i]),
    "asd_decoder_forward": (_i, [_vp, _i, _vp, _vp, _i64, _vp, _vp, _i, _i, _vp, _vp, _i64, _vp, _sz, _vp]),
    "asd_lm_head_partial": (_i, [_vp, _i64, _vp, _i64, _i, _i, _vp, _i, _i, _i, _i64, _f, _vp, _vp, _sz, _vp]),
    "asd_commit_step": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp, _i64, _vp, _i, _vp]),
    "asd_logprob_stats": (_i, [_vp, _i64, _vp, _i, _i, _vp, _vp]),
    "asd_mlp_packed_floats": (_sz, [_i, _i]),
    "asd_mlp_pack_weights": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp]),
    "asd_mlp_predict": (_i, [_vp, _i64, _vp, _i, _i, _i, _vp, _vp]),
    "asd_threshold_stop": (_i, [_vp, _vp, _i, _i, _vp, _vp]),
    "asd_bayes_adjust": (_i, [_vp, _i64, _d, _d, _i, _vp, _vp]),
    "asd_optimal_stopping": (_i, [_vp, _vp, _d, _i, _i, _i, _d, _d, _vp, _vp, _vp]),
    "asd_expected_cost": (_i, [_vp, _vp, _d, _vp, _i, _i, _vp, _vp]),
    "asd_lambda_sweep": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _d, _d, _vp, _vp, _vp, _vp]),
    "asd_derive_thresholds": (_i, [_vp, _vp, _i, _d, _vp, _vp]),
    "asd_predictor_stop": (_i, [_vp, _i64, _vp, _i, _vp, _i64, _i, _vp, _i, _i, _i, _i64, _d, _d, _vp, _vp, _d,
                                _i, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
}


# the TEST build's additional entry points (include/asd_hip.h, #ifdef ASD_TEST_HOOKS; lib/libasd_hip_test.so)
HOOK_SIGNATURES = {
    "asd_debug_verify_withhold": (_i, [_i]),
    "asd_debug_residual_groups": (_i, [_i]),
    "asd_debug_draft_groups": (_i, [_i]),
    "asd_debug_draft_withhold": (_i, [_i]),
    "asd_debug_force_linear_slices": (_i, [_i]),
    "asd_debug_linear_tall": (_i, [_i]),
}
TEST_LIB_PATH = os.path.join(_HERE, "lib", "libasd_hip_test.so")


class VerifyOptions(C.Structure):
    """asd_verify_options (include/asd_hip.h)."""
    _fields_ = [("inv_temperature", C.c_float), ("splits", C.c_int), ("threads", C.c_int), ("unroll", C.c_int),
                ("nontemporal", C.c_int)]


def verify_options(inv_temperature: float = 1.0, splits: int = 0, threads: int = 0, unroll: int = 0,
                   nontemporal: int = -1) -> VerifyOptions:
    return VerifyOptions(float(inv_temperature), int(splits), int(threads), int(unroll), int(nontemporal))


class Layer(C.Structure):
    """asd_layer_t"""
    _fields_ = [("ln1_w", _vp), ("qkv_w", _vp), ("qkv_b", _vp), ("o_w", _vp), ("ln2_w", _vp), ("gate_up_w", _vp), ("down_w", _vp),
                ("k_cache", _vp), ("vt_cache", _vp), ("weights_packed", _i)]


class DecoderShape(C.Structure):
    """asd_decoder_shape_t"""
    _fields_ = [("hidden", _i), ("heads", _i), ("kv_heads", _i), ("head_dim", _i), ("intermediate", _i), ("rms_eps", _f),
                ("inv_freq", _vp), ("t_max", _i)]


class AsdError(RuntimeError):
    """A libasd_hip.so entry point returned a negative asd_status."""

    def __init__(self, fn: str, status: int, text: str):
        super().__init__(f"{fn} failed: {text} (asd_status {status})")
        self.status = status


_lock = threading.Lock()
_lib = None
_test_lib = None
_override = threading.local()       # test suite only: the calling thread's wrappers go through the TEST build (use_test_library)


def load_test_library() -> C.CDLL:
    """dlopen lib/libasd_hip_test.so (the -DASD_TEST_HOOKS build: every product entry point + the asd_debug_* hooks), building
    it in-tree on first use.  Test infrastructure: nothing in the package calls this."""
    global _test_lib
    if _test_lib is not None:
        return _test_lib
    load_library()                      # torch's HIP runtime first (see below) -- and BEFORE taking the lock, which load_library takes too
    with _lock:
        if _test_lib is None:
            import importlib.util
            spec = importlib.util.spec_from_file_location("asd_amd_build", os.path.join(_HERE, "build.py"))
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
            mod.build_test_hooks()
            lib = C.CDLL(TEST_LIB_PATH)
            for name, (res, args) in list(SIGNATURES.items()) + list(HOOK_SIGNATURES.items()):
                fn = getattr(lib, name)
                fn.restype = res
                fn.argtypes = args
            _test_lib = lib
    return _test_lib


class use_test_library:
    """with use_test_library() as lib: ... -- inside the block THIS thread's load_library() returns the TEST build, so the
    package's wrappers launch its kernels and honour its asd_debug_* switches (called on `lib`).  Test infrastructure."""

    def __enter__(self):
        self._prev = getattr(_override, "lib", None)
        _override.lib = load_test_library()
        return _override.lib

    def __exit__(self, *exc):
        _override.lib = self._prev
        return False


def load_library() -> C.CDLL:
    """dlopen libasd_hip.so and attach the prototypes.  Raises if the library or a symbol is
    missing -- the caller must build it (`python adaptive-speculative-decoding_amd/build.py`)."""
    global _lib
    ov = getattr(_override, "lib", None)
    if ov is not None:
        return ov
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            _build_in_tree()
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: the HIP extension has not been built. "
                "Run `python adaptive-speculative-decoding_amd/build.py` (needs hipcc). "
                "There is no CPU fallback.")
        # The library and PyTorch must share ONE HIP runtime: device pointers and streams come from torch.
        # PyTorch-ROCm ships its own libamdhip64 (same SONAME as /opt/rocm's); whichever is mapped first
        # serves both, but only if torch's is first does torch use it too -- loaded the other way round the
        # process holds two runtimes and every call here fails with a HIP error on torch's pointers.
        import torch  # noqa: F401
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(lib, name)
            except AttributeError as e:  # pragma: no cover
                raise RuntimeError(f"{LIB_PATH} does not export {name}: rebuild it") from e
            fn.restype = res
            fn.argtypes = args
        _lib = lib
        return lib


def _build_in_tree() -> None:
    """First use on a fresh checkout: compile the library in-tree if a ROCm toolchain is present.
    Failure is not hidden -- load_library() then raises with the build instructions."""
    import importlib.util
    try:
        spec = importlib.util.spec_from_file_location("asd_amd_build", os.path.join(_HERE, "build.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        mod.build()
    except Exception as e:  # noqa: BLE001
        import sys
        print(f"[asd_amd] in-tree build of libasd_hip.so failed: {type(e).__name__}: {e}", file=sys.stderr)


def check(fn: str, status: int) -> None:
    if status != ASD_OK:
        text = load_library().asd_status_string(status).decode()
        raise AsdError(fn, status, text)
