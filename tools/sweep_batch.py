#!/usr/bin/env python3
"""Kernel time of the verify step against the batch size (K = 8, V = 152064, bf16): the plain kernel and the one-launch step, heuristic
geometry, hipGraph replays of back-to-back launches over rotating buffers (> 600 MB).  Dev tool; results go to gpurun_out/.

    python tools/sweep_batch.py [--batches 8,16,24,32,40,48,64,96,128,192,256] [--out gpurun_out/sweep_batch.json]
"""
import argparse
import json
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from asd_amd import kernels as K  # noqa: E402
from bench import N_STAGES, STAGE_COSTS, algorithmic_bytes, build_inputs, predictor_weights  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", default="8,16,24,32,40,48,64,96,128,192,256")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "sweep_batch.json"))
    ap.add_argument("--vocab", type=int, default=152064)
    ap.add_argument("--splits", default="0", help="forced split counts to time beside the heuristic (0), plain kernel only: e.g. 0,2,4")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    Kk, V = 8, a.vocab
    lib = K._lib()
    packed = K.pack_mlp_weights(*predictor_weights(np), device=dev)
    Cc = torch.tensor(STAGE_COSTS, dtype=torch.float64, device=dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    rows = []
    for B in [int(x) for x in a.batches.split(",")]:
        nbuf = max(3, math.ceil(640e6 / (B * Kk * V * 2)))
        ws, bufs = build_inputs(torch, K, B, Kk, V, nbuf, dev, 99)
        feat = torch.from_numpy((np.random.default_rng(7).standard_normal((B, 64)) * 0.3).astype(np.float32)).to(dev)
        ph = torch.ones((B, N_STAGES), dtype=torch.float64, device=dev)
        sc = torch.empty((B,), dtype=torch.float32, device=dev)
        ks = torch.empty((B,), dtype=torch.int32, device=dev)
        st_ = torch.empty((B,), dtype=torch.uint8, device=dev)

        def plain(buf):
            o = buf["out"]
            return lib.asd_verify_accept(buf["logits"].data_ptr(), 1, V, buf["tok"].data_ptr(), buf["lp_d"].data_ptr(), buf["u"].data_ptr(), B, Kk, V,
                                         o.lp_target.data_ptr(), o.accept.data_ptr(), o.n_acc.data_ptr(), o.accept_bits.data_ptr(),
                                         ws.buf.data_ptr(), ws.bytes, torch.cuda.current_stream().cuda_stream)

        def fused(buf):
            o = buf["out"]
            return lib.asd_verify_accept_fused(buf["logits"].data_ptr(), 1, V, buf["tok"].data_ptr(), buf["lp_d"].data_ptr(), buf["u"].data_ptr(), B, Kk, V,
                                               o.lp_target.data_ptr(), o.accept.data_ptr(), o.n_acc.data_ptr(), o.accept_bits.data_ptr(),
                                               ws.buf.data_ptr(), ws.bytes, feat.data_ptr(), 64, 5, packed.data_ptr(), 64, 32, 1, 100, 1.0, 1.0,
                                               ph.data_ptr(), Cc.data_ptr(), 1.0, N_STAGES, 0, 0, None, sc.data_ptr(), ks.data_ptr(), st_.data_ptr(),
                                               None, None, torch.cuda.current_stream().cuda_stream)
        nb = algorithmic_bytes(B, Kk, V)
        rec = {"batch": B, "rows": B * Kk, "algorithmic_bytes": nb}
        import ctypes
        from asd_amd._binding import verify_options
        forced = []
        for S in [int(x) for x in a.splits.split(",") if int(x) > 0]:
            opt = verify_options(1.0, S, 0, 0, -1)

            def plain_s(buf, opt=opt):
                o = buf["out"]
                return lib.asd_verify_accept_ex(buf["logits"].data_ptr(), 1, V, buf["tok"].data_ptr(), buf["lp_d"].data_ptr(), buf["u"].data_ptr(), B, Kk, V,
                                                o.lp_target.data_ptr(), o.accept.data_ptr(), o.n_acc.data_ptr(), o.accept_bits.data_ptr(),
                                                ws.buf.data_ptr(), ws.bytes, ctypes.addressof(opt), torch.cuda.current_stream().cuda_stream)
            if plain_s(bufs[0]) == 0:
                forced.append((f"plain_S{S}", plain_s))
        for name, fn in [("plain", plain), ("one_launch", fused)] + forced:
            for i in range(max(400, int(60e-3 / (nb / 4.0e12)))):     # settle: clocks / memory power state
                assert fn(bufs[i % nbuf]) == 0
            torch.cuda.synchronize()
            per = 24
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for i in range(per):
                    fn(bufs[i % nbuf])
            g.replay()
            torch.cuda.synchronize()
            runs = []
            for _ in range(5):
                e0.record()
                for _ in range(10):
                    g.replay()
                e1.record()
                torch.cuda.synchronize()
                runs.append(e0.elapsed_time(e1) / (10 * per) * 1e3)
            del g
            us = sorted(runs)[len(runs) // 2]
            rec[name + "_us"] = round(us, 3)
            rec[name + "_TBps"] = round(nb / us / 1e6, 3)
            rec[name + "_frac_of_8TBps"] = round(nb / us / 1e6 / 8.0, 4)
        assert ws.status() == 0
        print(rec, flush=True)
        rows.append(rec)
        del bufs, ws
        torch.cuda.empty_cache()
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump({"what": "verify step kernel time against the batch size (K = 8, V = 152064, bf16, heuristic geometry); median of 5 runs of 240 "
                       "graph-replayed launches", "rows": rows}, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
