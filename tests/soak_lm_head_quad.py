#!/usr/bin/env python3
"""Soak (run by hand on a GPU box, not collected by pytest): 24 random shapes with 257 <= M < 1100 -- the 4-wave
k_lm_head_quad path: ragged rows / columns, 1-8 superstages, K up to 32 -- against the f64 oracle, through the helpers of
tests/test_gpu_lm_head.py.

    python tests/soak_lm_head_quad.py [cases] [m_lo] [m_hi] [seed]        e.g.  200 257 289 7: the tall row block (256 < M <= 288)
"""
import sys, os, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import importlib.util
spec = importlib.util.spec_from_file_location("tl", os.path.join(sys.path[0], "tests", "test_gpu_lm_head.py"))
tl = importlib.util.module_from_spec(spec); spec.loader.exec_module(tl)
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
m_lo = int(sys.argv[2]) if len(sys.argv) > 2 else 257
m_hi = int(sys.argv[3]) if len(sys.argv) > 3 else 1100
rng = np.random.default_rng(int(sys.argv[4]) if len(sys.argv) > 4 else 2026)
n = 0
for it in range(cases):
    K = int(rng.integers(1, 33))
    M = int(rng.integers(m_lo, m_hi))
    B = max(1, M // K)
    while B * K < m_lo and B * K + K < m_hi:       # M // K * K can fall below the range asked for
        B += 1
    D = 64 * int(rng.integers(1, 9))
    V = int(rng.choice([int(rng.integers(5, 3000)), int(rng.integers(3000, 70000))], p=[0.8, 0.2]))
    case = tl.make_case(B, K, D, V, seed=1000 + it)
    tl.check(tl.run_gpu(case), case["ref"])
    n += 1
    print(it, B, K, D, V, "ok", flush=True)
print("fuzz cases passed:", n)
