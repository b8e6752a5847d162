"""Soak of asd_linear_ex on random shapes (run by hand on a GPU box, not collected):
    python tests/soak_linear.py [cases] [seed]
M in [1, 640], N a multiple of 4 in [4, 12288], D a multiple of 64 in [64, 8192]; bias / residual / strided operands / f16 at random;
every row-count regime (stream-shaped, 8-wave tile with a short last block, the 288-row block, the 4-wave kernel) and random
forced slice counts.  Bound as in tests/test_gpu_linear.py: |y - f64 ref| <= 2^-8 |ref| + 2e-4 (bf16), 2^-11 |ref| + 2e-4 (f16)."""
import sys
from importlib import import_module
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
K_ = import_module("adaptive-speculative-decoding_amd.kernels")


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 150
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = np.random.default_rng(seed)
    lib = K_.test_hooks().__enter__()          # the TEST build of the library for the whole program (asd_debug_force_linear_slices)
    ws = K_.LinearWorkspace("cuda")
    worst = 0.0
    for c in range(cases):
        regime = rng.integers(0, 5)
        M = int([rng.integers(1, 65), rng.integers(65, 257), rng.integers(257, 289), rng.integers(289, 641), rng.integers(1, 641)][regime])
        N = int(rng.integers(1, 3073)) * 4
        D = int(rng.integers(1, 129)) * 64
        if M * N * D > 3.0e10:
            D = 64 * max(1, int(3.0e10 / (M * N)) // 64)
        dtype = torch.bfloat16 if rng.random() < 0.7 else torch.float16
        pad_x, pad_w = (int(rng.integers(0, 3)) * 8 for _ in range(2))
        g = torch.Generator(device="cuda").manual_seed(int(rng.integers(0, 2 ** 31)))
        x = torch.randn(M, D + pad_x, generator=g, device="cuda").to(dtype)[:, :D]
        w = (torch.randn(N, D + pad_w, generator=g, device="cuda") / D ** 0.5).to(dtype)[:, :D]
        b = torch.randn(N, generator=g, device="cuda").to(dtype) if rng.random() < 0.5 else None
        res = torch.randn(M, N, generator=g, device="cuda").to(dtype) if rng.random() < 0.5 else None
        force = int(rng.integers(0, 4))
        k = 0 if force else int(rng.integers(1, min(32, D // 64) + 1))
        lib.asd_debug_force_linear_slices(k)
        ws.buf = torch.empty(max(1, k) * M * N * 4 + (1 << 20), dtype=torch.uint8, device="cuda") if k else ws.buf
        try:
            y = K_.linear(x, w, b, workspace=ws, residual=res)
        finally:
            lib.asd_debug_force_linear_slices(0)
        ref = x.double() @ w.double().T
        if b is not None:
            ref = ref + b.double()
        if res is not None:
            ref = ref + res.double()
        rel = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
        excess = ((y.double() - ref).abs() - (rel * ref.abs() + 2e-4)).max().item()
        worst = max(worst, excess)
        assert excess <= 0.0, f"case {c}: M={M} N={N} D={D} {dtype} slices={k} bias={b is not None} res={res is not None}: excess {excess:.3e}"
        if c % 25 == 0:
            print(f"case {c}: M={M} N={N} D={D} {str(dtype)[6:]} forced_slices={k} ok", flush=True)
    print(f"soak_linear: {cases} cases passed (seed {seed}); worst excess over the bound {worst:.3e}")


if __name__ == "__main__":
    main()
