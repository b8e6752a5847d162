"""The C-ABI shared library loads on a CPU-only box and exports every symbol include/asd_hip.h
declares; the ctypes prototypes cover exactly that set.  No compute call is made here except
the host-only entry points (asd_version, asd_status_string, asd_mlp_pack*, asd_derive_thresholds)."""
import ctypes as C
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(hooks: bool = False):
    """Entry points include/asd_hip.h declares: the product's (outside `#ifdef ASD_TEST_HOOKS`) or the test hooks (inside)."""
    text = open(os.path.join(ROOT, "include", "asd_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    inside = "".join(re.findall(r"#ifdef ASD_TEST_HOOKS(.*?)#endif", text, flags=re.S))
    outside = re.sub(r"#ifdef ASD_TEST_HOOKS.*?#endif", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(asd_[a-z0-9_]+)\s*\(", inside if hooks else outside)))


def _exported(path):
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
    return sorted({ln.split()[-1] for ln in out.splitlines() if ln.split() and ln.split()[-1].startswith("asd_")})


def test_header_symbols_exported_and_bound():
    from asd_amd import _binding
    lib = _binding.load_library()
    names = _declared()
    assert len(names) >= 19
    for n in names:
        assert hasattr(lib, n), f"{n} not exported by libasd_hip.so"
    assert sorted(_binding.SIGNATURES) == names


def test_the_product_library_exports_no_test_hook():
    """The asd_debug_* switches are process-global (not thread-safe next to a 100-thread caller, src/serving/pipeline.py:83):
    they live in the TEST build of the library only.  libasd_hip.so exports exactly the header's product entry points."""
    from asd_amd import _binding
    _binding.load_library()
    exported = _exported(_binding.LIB_PATH)
    assert not [n for n in exported if n.startswith("asd_debug_")], exported
    assert exported == _declared()
    hooks = _declared(hooks=True)
    assert hooks and all(n.startswith("asd_debug_") for n in hooks) and sorted(_binding.HOOK_SIGNATURES) == hooks
    # ... and the test build (hipcc cross-compiles it here) exports the product's entry points plus exactly those hooks
    test_lib = _binding.load_test_library()
    assert _exported(_binding.TEST_LIB_PATH) == sorted(_declared() + hooks)
    for n in hooks:
        assert hasattr(test_lib, n)


def test_version_and_status_strings():
    from asd_amd import _binding
    lib = _binding.load_library()
    assert lib.asd_version() == 300       # 0.3.0: every hand-off workspace starts with a status block; the asd_debug_* hooks left the product
    assert lib.asd_status_string(0) == b"ok"
    assert b"workspace" in lib.asd_status_string(-3)
    assert lib.asd_verify_accept_workspace_bytes(32, 8, 152064, 1) % 256 == 0
    assert lib.asd_mlp_packed_floats(64, 32) == 2113


def test_host_entry_points_match_goldens(golden):
    """asd_derive_thresholds and asd_mlp_pack_weights run on the host: bit-exact vs the goldens."""
    from asd_amd import kernels as K
    for row in golden.json("thresholds.json")["rows"]:
        theta, _ = K.derive_thresholds(row["q"], row["c"], row["lam"])
        assert theta.tolist() == row["theta"]
    from asd_amd import _binding
    lib = _binding.load_library()
    g = golden.npz("predictor.npz")
    packed = np.empty(2113, np.float32)
    vp = lambda a: np.ascontiguousarray(a, dtype=np.float32).ctypes.data_as(C.c_void_p)
    w1, b1, w2, b2 = (np.ascontiguousarray(g[k], dtype=np.float32) for k in ("w1", "b1", "w2", "b2"))
    assert lib.asd_mlp_pack_weights(w1.ctypes.data_as(C.c_void_p), b1.ctypes.data_as(C.c_void_p),
                                    w2.ctypes.data_as(C.c_void_p), b2.ctypes.data_as(C.c_void_p), 64, 32,
                                    packed.ctypes.data_as(C.c_void_p)) == 0
    assert np.array_equal(packed[:64 * 32].reshape(64, 32), w1.T)
    assert np.array_equal(packed[2048:2080], b1) and np.array_equal(packed[2080:2112], w2[0]) and packed[2112] == b2[0]


def test_product_fails_loudly_without_gpu():
    import pytest
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from asd_amd import backend
    backend.set_backend(None)
    from asd_amd.algorithms import optimal_stopping_rule
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        optimal_stopping_rule([0.5], [1.0], 1.0)


def test_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "adaptive-speculative-decoding_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f
                assert "libasd_oracle" not in text, f


def test_linear_plan_is_sane_without_a_gpu():
    """X3: the reduction-slice plan of asd_linear is host arithmetic (a device with no visible GPU counts as 256 CUs): at least
    one slice, never more than the reduction has superstages, none for a grid that already fills the CUs, and a workspace
    that holds exactly the plan's partials."""
    import ctypes as C
    from asd_amd import _binding as B
    lib = B.load_library()
    for (M, N, D) in [(1, 3584, 3584), (32, 3584, 3584), (32, 152064, 3584), (64, 512, 64), (99, 8192, 29568), (207, 59136, 8192),
                      (288, 5120, 27648), (288, 55296, 5120), (992, 3584, 3584), (1024, 152064, 8192)]:
        k = lib.asd_linear_slices(M, N, D)
        assert 1 <= k <= min(32, D // 64), (M, N, D, k)
        need = lib.asd_linear_workspace_bytes(M, N, D)
        assert need >= (k * M * N * 4 if k > 1 else 0) and need <= k * M * N * 4 + 1024
    assert lib.asd_linear_slices(32, 59136, 8192) == 1            # 231 column blocks on 256 CUs: one round, no slicing
    assert lib.asd_linear_slices(32, 3584, 3584) > 1              # 14 column blocks
    assert lib.asd_linear_slices(32, 3584, 100) == 0 and lib.asd_linear_workspace_bytes(32, 3584, 100) == 0   # D % 64


def test_the_test_library_can_be_the_first_thing_a_fresh_process_loads():
    """load_test_library() loads the product library first; it must not do so while holding the lock load_library() takes (a
    stand-alone soak script whose first call was `kernels.test_hooks()` waited for ever; the suite never saw it because its
    fixtures load the product library first)."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); from asd_amd import _binding as B; lib = B.use_test_library().__enter__(); "
            "print(lib.asd_version(), hasattr(lib, 'asd_debug_draft_groups'))" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.split() == ["300", "True"]
