#!/usr/bin/env python3
"""Geometry sweep of asd_verify_accept_ex (asd_verify_options) on one GPU (dev tool; results go to gpurun_out/).

    python tools/sweep_verify.py [--workload c3] [--reps 60] [--out gpurun_out/sweep.json]

Per configuration: `reps` back-to-back launches rotating through > 600 MB of logits buffers,
bracketed by two HIP events; reports us per launch (includes the ~1.5 us inter-kernel gap) and
the algorithmic GB/s.
"""
import argparse
import itertools
import json
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from asd_amd import kernels as K  # noqa: E402
from bench import WORKLOADS, algorithmic_bytes, build_inputs  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c3")
    ap.add_argument("--reps", type=int, default=60)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "sweep.json"))
    ap.add_argument("--threads", default="256,512,1024")
    ap.add_argument("--unroll", default="2,4,8")
    ap.add_argument("--splits", default="1,2,3,4,6,8,9,11,12,16,18")
    ap.add_argument("--nt", default="0,1")
    a = ap.parse_args()
    B, Kk, V, _ = WORKLOADS[a.workload]
    dev = torch.device("cuda", 0)
    nbuf = max(3, math.ceil(640e6 / (B * Kk * V * 2)))
    ws, bufs = build_inputs(torch, K, B, Kk, V, nbuf, dev, 1234)
    lib = K._lib()
    st = torch.cuda.current_stream().cuda_stream
    nbytes = algorithmic_bytes(B, Kk, V)
    rows = []

    import ctypes
    from asd_amd._binding import verify_options

    def launch(buf, g):
        o = buf["out"]
        opt = verify_options(1.0, *g)
        return lib.asd_verify_accept_ex(buf["logits"].data_ptr(), 1, V, buf["tok"].data_ptr(), buf["lp_d"].data_ptr(),
                                        buf["u"].data_ptr(), B, Kk, V, o.lp_target.data_ptr(), o.accept.data_ptr(),
                                        o.n_acc.data_ptr(), o.accept_bits.data_ptr(), ws.buf.data_ptr(), ws.bytes,
                                        ctypes.addressof(opt), torch.cuda.current_stream().cuda_stream)

    grid = itertools.product([int(x) for x in a.threads.split(",")], [int(x) for x in a.unroll.split(",")],
                             [int(x) for x in a.splits.split(",")], [int(x) for x in a.nt.split(",")])
    for threads, unroll, splits, nt in grid:
        g = (splits, threads, unroll, nt)
        if launch(bufs[0], g) != 0:
            continue
        for i in range(10):
            launch(bufs[i % nbuf], g)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        # a B = 8 launch is shorter than a ctypes call: time replays of a hipGraph of `per` launches (round 3)
        per, graph = 24, None
        try:
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                for i in range(per):
                    launch(bufs[i % nbuf], g)
            graph.replay()
            torch.cuda.synchronize()
        except Exception:  # noqa: BLE001
            graph = None
        best = None
        for _ in range(3):
            e0.record()
            if graph is not None:
                for _r in range(max(1, a.reps // per)):
                    graph.replay()
                n_l = max(1, a.reps // per) * per
            else:
                for i in range(a.reps):
                    launch(bufs[i % nbuf], g)
                n_l = a.reps
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / n_l
            best = us if best is None else min(best, us)
        del graph
        rows.append(dict(threads=threads, unroll=unroll, splits=splits, nt=nt, us=best, gbs=nbytes / best / 1e3))
        print(f"T={threads:5d} U={unroll} S={splits:3d} nt={nt}  {best:8.2f} us  {nbytes / best / 1e3:8.1f} GB/s", flush=True)
    rows.sort(key=lambda r: r["us"])
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, "w") as f:
        json.dump(dict(workload=a.workload, B=B, K=Kk, V=V, bytes=nbytes, rows=rows), f, indent=1)
    print("best:", rows[:5])


if __name__ == "__main__":
    main()
