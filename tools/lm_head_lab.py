#!/usr/bin/env python3
"""Dev tool for the N2 GEMM kernel: builds variants of csrc/lm_head_verify.hip with -D flags into
adaptive-speculative-decoding_amd/lib/lab/ (here, no GPU needed) and times them on the GPU box.

    python tools/lm_head_lab.py --build name:-DFLAG=1,-DOTHER=2 [name2:...]     # in the container
    python tools/lm_head_lab.py --run [--shapes 7b,72b]                         # on the GPU box

Variants that drop loads or math give wrong results on purpose; only their time is of interest.
"""
import argparse
import ctypes as C
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = os.path.join(ROOT, "adaptive-speculative-decoding_amd")
LAB = os.path.join(PKG, "lib", "lab")
SHAPES = {"7b": 3584, "32b": 5120, "72b": 8192}


def build(specs):
    os.makedirs(LAB, exist_ok=True)
    for spec in specs:
        name, _, flags = spec.partition(":")
        out = os.path.join(LAB, f"lm_head_{name}.so")
        cmd = ["hipcc", "-std=c++17", "-O3", "--offload-arch=gfx950", "-fPIC", "-shared", "-fno-fast-math",
               "-ffp-contract=off", f"-I{ROOT}/include", f"-I{PKG}/csrc", *[f for f in flags.split(",") if f],
               os.path.join(PKG, "csrc", "lm_head_verify.hip"), os.path.join(PKG, "csrc", "api.hip"), "-o", out]
        subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
        print("built", out)


def run(shapes, reps, out_path):
    import torch

    B, Kk, V = 32, 8, 152064
    M = B * Kk
    res = []
    libs = sorted(glob.glob(os.path.join(LAB, "lm_head_*.so")))
    for name in shapes:
        D = SHAPES[name]
        g = torch.Generator(device="cuda").manual_seed(1)
        h = torch.randn((M, D), device="cuda", generator=g).to(torch.bfloat16)
        w = (torch.randn((V, D), device="cuda", generator=g) * (3.0 / D ** 0.5)).to(torch.bfloat16)
        tok = torch.randint(0, V, (B, Kk), device="cuda", dtype=torch.int32)
        lp_d = -torch.rand((B, Kk), device="cuda")
        u = torch.rand((B, Kk), device="cuda")
        lp = torch.empty((B, Kk), device="cuda")
        acc = torch.empty((B, Kk), dtype=torch.uint8, device="cuda")
        n = torch.empty((B,), dtype=torch.int32, device="cuda")
        for path in libs:
            lib = C.CDLL(path)
            lib.asd_lm_head_verify_workspace_bytes.restype = C.c_size_t
            lib.asd_lm_head_verify_workspace_bytes.argtypes = [C.c_int] * 3
            lib.asd_lm_head_verify.restype = C.c_int
            lib.asd_lm_head_verify.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p,
                                               C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p,
                                               C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
            nb = lib.asd_lm_head_verify_workspace_bytes(B, Kk, V)
            ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
            st = torch.cuda.current_stream().cuda_stream

            def call():
                rc = lib.asd_lm_head_verify(h.data_ptr(), D, w.data_ptr(), D, 1, D, tok.data_ptr(), lp_d.data_ptr(),
                                            u.data_ptr(), B, Kk, V, 1.0, lp.data_ptr(), acc.data_ptr(), n.data_ptr(), None,
                                            ws.data_ptr(), nb, st)
                assert rc == 0, rc

            for _ in range(3):
                call()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            best = 1e30
            for _ in range(3):
                e0.record()
                for _ in range(reps):
                    call()
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
            row = dict(shape=name, variant=os.path.basename(path)[8:-3], us=best, tflops=2.0 * M * D * V / best / 1e6,
                       w_gbs=V * D * 2 / best / 1e3)
            res.append(row)
            print(json.dumps(row), flush=True)
    os.makedirs(os.path.dirname(out_path), exist_ok=True)
    with open(out_path, "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", nargs="*")
    ap.add_argument("--run", action="store_true")
    ap.add_argument("--shapes", default="7b,72b")
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "lm_head_lab.json"))
    a = ap.parse_args()
    if a.build:
        build(a.build)
    if a.run:
        run(a.shapes.split(","), a.reps, a.out)
