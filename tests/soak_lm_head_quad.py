#!/usr/bin/env python3
"""Soak (run by hand on a GPU box, not collected by pytest): 24 random shapes with 257 <= M < 1100 -- the 4-wave
k_lm_head_quad path: ragged rows / columns, 1-8 superstages, K up to 32 -- against the f64 oracle, through the helpers of
tests/test_gpu_lm_head.py.

    python tests/soak_lm_head_quad.py
"""
import sys, os, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import importlib.util
spec = importlib.util.spec_from_file_location("tl", os.path.join(sys.path[0], "tests", "test_gpu_lm_head.py"))
tl = importlib.util.module_from_spec(spec); spec.loader.exec_module(tl)
rng = np.random.default_rng(2026)
n = 0
for it in range(24):
    K = int(rng.integers(1, 33))
    M = int(rng.integers(257, 1100))
    B = max(1, M // K)
    D = 64 * int(rng.integers(1, 9))
    V = int(rng.integers(5, 3000))
    case = tl.make_case(B, K, D, V, seed=1000 + it)
    tl.check(tl.run_gpu(case), case["ref"])
    n += 1
    print(it, B, K, D, V, "ok", flush=True)
print("fuzz cases passed:", n)
