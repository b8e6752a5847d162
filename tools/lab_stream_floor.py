#!/usr/bin/env python3
"""Lab: the floor of the verify step's stream -- a kernel that only READS the [B*K rows][V] bf16 logits with k_verify's launch
shape and load instructions (tools/lab_stream_floor.hip), timed like bench.py times k_verify: back-to-back launches over > 600 MB
of rotating buffers, interleaved with the real plain kernel in one process.

    python tools/lab_stream_floor.py [c3|c5] [--out gpurun_out/lab_stream_floor.json]
"""
import argparse
import ctypes as C
import json
import math
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from bench import WORKLOADS, algorithmic_bytes, build_inputs  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("workload", nargs="?", default="c3")
    ap.add_argument("--rounds", type=int, default=6)
    ap.add_argument("--reps", type=int, default=300)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "lab_stream_floor.json"))
    a = ap.parse_args()
    out_dir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    so = os.path.join(out_dir, "liblab_stream_floor.so")
    subprocess.check_call(["hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-shared", os.path.join(ROOT, "tools", "lab_stream_floor.hip"), "-o", so])
    from asd_amd import kernels as K
    lab = C.CDLL(so)
    lab.lab_stream.restype = C.c_int
    lab.lab_stream.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    B, Kk, V, _ = WORKLOADS[a.workload]
    dev = torch.device("cuda", 0)
    nbuf = max(3, math.ceil(640e6 / (B * Kk * V * 2)))
    ws, bufs = build_inputs(torch, K, B, Kk, V, nbuf, dev, 1234)
    sink = torch.zeros((B * Kk * 16 * 1024,), dtype=torch.int32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    rows, row_bytes = B * Kk, V * 2

    def plain(j):
        K.verify_accept(bufs[j % nbuf]["logits"], bufs[j % nbuf]["tok"], bufs[j % nbuf]["lp_d"], bufs[j % nbuf]["u"], ws, bufs[j % nbuf]["out"])

    def lab_fn(mode, threads, depth, nt, splits):
        def f(j):
            rc = lab.lab_stream(bufs[j % nbuf]["logits"].data_ptr(), rows, row_bytes, row_bytes, mode, threads, depth, nt, splits, sink.data_ptr(), st)
            assert rc == 0, rc
        return f
    fns = {"k_verify plain": plain,
           "read only: dynamic 512 x 3 KiB nt (k_verify's loop)": lab_fn(1, 512, 3, 1, 1),
           "read only: dynamic 512 x 3 KiB": lab_fn(1, 512, 3, 0, 1),
           "read only: dynamic 512 x 4 KiB nt": lab_fn(1, 512, 4, 1, 1),
           "read only: dynamic 1024 x 2 KiB nt": lab_fn(1, 1024, 2, 1, 1),
           "read only: static 512 x depth 6 nt": lab_fn(0, 512, 6, 1, 1),
           "read only: static 512 x depth 6": lab_fn(0, 512, 6, 0, 1),
           "read only: static 512 x depth 8 nt": lab_fn(0, 512, 8, 1, 1),
           "read only: static 1024 x depth 4 nt": lab_fn(0, 1024, 4, 1, 1),
           "read only: static 512 x depth 6 nt, 2 slices per row": lab_fn(0, 512, 6, 1, 2),
           "read only: static 256 x depth 8 nt, 4 slices per row": lab_fn(0, 256, 8, 1, 4)}
    for f in fns.values():
        for j in range(300):
            f(j)
    torch.cuda.synchronize()
    res = {k: [] for k in fns}
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(a.rounds):
        for name, f in fns.items():
            for j in range(40):
                f(j)
            e0.record()
            for j in range(a.reps):
                f(j)
            e1.record()
            torch.cuda.synchronize()
            res[name].append(e0.elapsed_time(e1) * 1000.0 / a.reps)
    nbytes = algorithmic_bytes(B, Kk, V)
    out = {"workload": a.workload, "bytes": nbytes, "us": {k: {"median": float(np.median(v)), "min": float(np.min(v)), "TBs": nbytes / (float(np.median(v)) * 1e-6) / 1e12} for k, v in res.items()}}
    for k, v in out["us"].items():
        print(f"{k:60s} {v['median']:7.2f} us   {v['TBs']:.2f} TB/s")
    with open(a.out, "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
