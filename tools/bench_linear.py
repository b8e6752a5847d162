"""X3: asd_linear beside torch's F.linear (hipBLASLt / rocBLAS) on the decoder-layer projections of the Qwen2.5 shapes at the
row counts of the token-level loop: M = 32 (draft, one token per sequence), 208 / 288 (verify tiers, K + 1 positions).
Weights rotate over enough copies to exceed the 256 MB MALL, so the figures are HBM figures.
    python tools/bench_linear.py [--out gpurun_out/linear.json] [--models 7b,32b,72b] [--rows 32,288]"""
import argparse
import json
import sys
from importlib import import_module
from pathlib import Path

import torch
import torch.nn.functional as F

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
K_ = import_module("adaptive-speculative-decoding_amd.kernels")
SL = import_module("adaptive-speculative-decoding_amd.serving.synthetic_lm")


def time_us(fn, n_rot, replays=8):
    """n_rot calls (one per rotating weight copy) captured in ONE hipGraph, replayed: no per-call host overhead in the figure."""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for i in range(n_rot):
            fn(i)
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for i in range(n_rot):
            fn(i)
    graph.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(replays):
        graph.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (replays * n_rot)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--models", default="7b,32b,72b")
    ap.add_argument("--rows", default="32,288")
    ap.add_argument("--force-slices", type=int, default=0)
    ap.add_argument("--no-tall", action="store_true", help="256 < M <= 288 as 256 + 32 rows instead of one 288-row block")
    ap.add_argument("--tall-mode", type=int, default=None, help="asd_debug_linear_tall: 0 off, 1 for 256 < M <= 288, 2 also for 192 < M <= 256")
    ap.add_argument("--no-torch", action="store_true")
    a = ap.parse_args()
    lib = K_.test_hooks().__enter__()          # the TEST build of the library for the whole program (asd_debug_* switches)
    lib.asd_debug_force_linear_slices(a.force_slices)
    if a.tall_mode is not None:
        lib.asd_debug_linear_tall(a.tall_mode)
    elif a.no_tall:
        lib.asd_debug_linear_tall(0)
    ws = K_.LinearWorkspace("cuda")
    res = []
    for name in a.models.split(","):
        s = SL.QWEN25_SHAPES[name]
        kv = s.kv_heads * s.head_dim
        mats = {"qkv": (s.hidden + 2 * kv, s.hidden), "o": (s.hidden, s.hidden), "gate_up": (2 * s.intermediate, s.hidden),
                "down": (s.hidden, s.intermediate)}
        for mname, (N, D) in mats.items():
            wbytes = N * D * 2
            n_rot = max(2, min(8, (600 << 20) // wbytes + 1))
            W = [torch.randn(N, D, device="cuda", dtype=torch.bfloat16) * D ** -0.5 for _ in range(n_rot)]
            for M in [int(v) for v in a.rows.split(",")]:
                x = torch.randn(M, D, device="cuda", dtype=torch.bfloat16)
                out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
                t_asd = time_us(lambda i: K_.linear(x, W[i], workspace=ws, out=out), n_rot)
                t_torch = float("nan") if a.no_torch else time_us(lambda i: F.linear(x, W[i]), n_rot)
                r = {"model": name, "matrix": mname, "M": M, "N": N, "D": D, "slices": int(lib.asd_linear_slices(M, N, D)),
                     "asd_us": round(t_asd, 2), "torch_us": round(t_torch, 2), "asd_TBps": round(wbytes / t_asd / 1e6, 3),
                     "torch_TBps": round(wbytes / t_torch / 1e6, 3), "asd_TFLOPs": round(2.0 * M * N * D / t_asd / 1e6, 1)}
                res.append(r)
                print(r, flush=True)
            del W
            torch.cuda.empty_cache()
    if a.out:
        Path(a.out).write_text(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
