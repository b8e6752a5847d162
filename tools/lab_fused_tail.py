#!/usr/bin/env python3
"""Lab: what does each piece of the one-launch step's tail cost?  Builds variants of verify_accept.hip with -DASD_LAB=n into
gpurun_out/ (the shipped library never contains them), then times the FUSED instantiation of every variant back to back,
interleaved in ONE process (rounds x reps launches each, rotating > 600 MB of logits), next to the plain kernel.

    python tools/lab_fused_tail.py [c3|c5|c2] [--variants 0,1,2] [--rounds 8] [--reps 300] [--out gpurun_out/lab_fused_tail.json]

  variant 0   the product code
          1   no in-kernel epilogue (the finisher stops behind the hand-off)
          2   no epilogue and no wait for the sibling rows
  further variants: see the ASD_LAB blocks in csrc/verify_accept.hip
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

from bench import N_STAGES, STAGE_COSTS, WORKLOADS, build_inputs, predictor_weights  # noqa: E402


# variants >= 20 are the product code (ASD_LAB = 0) under other build switches
SWITCHES = {20: ["-DASD_PARAMS_EARLY=1"], 21: ["-DASD_PARAMS_EARLY=0"], 22: ["-fno-slp-vectorize"]}


def build_variant(n: int) -> str:
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    lib_path = os.path.join(out, f"libasd_hip_lab{n}.so")
    csrc = os.path.join(ROOT, "adaptive-speculative-decoding_amd", "csrc")
    subprocess.check_call(["hipcc", "-std=c++17", "-O3", "--offload-arch=gfx950", "-fPIC", "-shared", f"-DASD_LAB={0 if n >= 20 else n}", *SWITCHES.get(n, []),
                           "-fno-fast-math", "-ffp-contract=off", "-mllvm", "-amdgpu-kernarg-preload-count=11", f"-I{ROOT}/include", f"-I{csrc}",
                           os.path.join(csrc, "verify_accept.hip"), os.path.join(csrc, "api.hip"), os.path.join(csrc, "predictor.hip"),
                           "-o", lib_path])
    return lib_path


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("workload", nargs="?", default="c3")
    ap.add_argument("--variants", default="0,1,2")
    ap.add_argument("--rounds", type=int, default=8)
    ap.add_argument("--reps", type=int, default=300)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "lab_fused_tail.json"))
    ap.add_argument("--also-256x128", action="store_true", help="time the first variant's one-launch step with the 256 -> 128 -> 1 predictor too")
    ap.add_argument("--extra-libs", default="", help="name=path,... : prebuilt libraries (e.g. an older commit's build) timed in the same process")
    a = ap.parse_args()
    from asd_amd import kernels as K
    from asd_amd._binding import SIGNATURES
    B, Kk, V, _ = WORKLOADS[a.workload]
    dev = torch.device("cuda", 0)
    nbuf = max(3, math.ceil(640e6 / (B * Kk * V * 2)))
    ws, bufs = build_inputs(torch, K, B, Kk, V, nbuf, dev, 1234)
    packed = K.pack_mlp_weights(*predictor_weights(np), device=dev)
    feat = torch.from_numpy((np.random.default_rng(7).standard_normal((B, 64)) * 0.3).astype(np.float32)).to(dev)
    Cc = torch.tensor(STAGE_COSTS, dtype=torch.float64, device=dev)
    ph = torch.ones((B, N_STAGES), dtype=torch.float64, device=dev)
    score = torch.empty((B,), dtype=torch.float32, device=dev)
    ks = torch.empty((B,), dtype=torch.int32, device=dev)
    stp = torch.empty((B,), dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    variants = [int(x) for x in a.variants.split(",")]
    fns = {}
    for n in variants:
        lib = C.CDLL(build_variant(n))
        ff = lib.asd_verify_accept_fused_ex
        ff.restype, ff.argtypes = SIGNATURES["asd_verify_accept_fused_ex"]
        pl = lib.asd_verify_accept_ex
        pl.restype, pl.argtypes = SIGNATURES["asd_verify_accept_ex"]
        fns[f"fused/{n}"] = ("fused", ff)
        if n == variants[0]:
            fns["plain"] = ("plain", pl)
        elif n in (4, 8, 9, 10) or n >= 20:
            fns[f"plain/{n}"] = ("plain", pl)

    big = None
    if a.also_256x128:
        rngb = np.random.default_rng(3)
        pk = K.pack_mlp_weights((rngb.standard_normal((128, 256)) / 16).astype(np.float32), np.zeros(128, np.float32),
                                (rngb.standard_normal((1, 128)) / 8).astype(np.float32), np.zeros(1, np.float32), device=dev)
        fb = torch.from_numpy((rngb.standard_normal((B, 256)) * 0.3).astype(np.float32)).to(dev)
        big = (pk, fb)
        fns["fused 256x128"] = ("fused256", fns[f"fused/{variants[0]}"][1])
    for item in [x for x in a.extra_libs.split(",") if x]:
        name, path = item.split("=")
        lib = C.CDLL(path if os.path.isabs(path) else os.path.join(ROOT, path))
        ff = lib.asd_verify_accept_fused_ex
        ff.restype, ff.argtypes = SIGNATURES["asd_verify_accept_fused_ex"]
        pl = lib.asd_verify_accept_ex
        pl.restype, pl.argtypes = SIGNATURES["asd_verify_accept_ex"]
        fns[f"fused@{name}"] = ("fused", ff)
        fns[f"plain@{name}"] = ("plain", pl)

    def launch(kind, fn, j):
        bj = bufs[j % nbuf]
        o = bj["out"]
        if kind == "plain":
            return fn(bj["logits"].data_ptr(), 1, V, bj["tok"].data_ptr(), bj["lp_d"].data_ptr(), bj["u"].data_ptr(), B, Kk, V,
                      o.lp_target.data_ptr(), o.accept.data_ptr(), o.n_acc.data_ptr(), o.accept_bits.data_ptr(), ws.buf.data_ptr(),
                      ws.bytes, None, st)
        if kind == "fused256":
            return fn(bj["logits"].data_ptr(), 1, V, bj["tok"].data_ptr(), bj["lp_d"].data_ptr(), bj["u"].data_ptr(), B, Kk, V,
                      o.lp_target.data_ptr(), o.accept.data_ptr(), o.n_acc.data_ptr(), o.accept_bits.data_ptr(), ws.buf.data_ptr(), ws.bytes,
                      big[1].data_ptr(), 256, 5, big[0].data_ptr(), 256, 128, 1, 100, 1.0, 1.0, ph.data_ptr(), Cc.data_ptr(), 1.0, N_STAGES, 0, 0,
                      None, score.data_ptr(), ks.data_ptr(), stp.data_ptr(), None, None, None, st)
        return fn(bj["logits"].data_ptr(), 1, V, bj["tok"].data_ptr(), bj["lp_d"].data_ptr(), bj["u"].data_ptr(), B, Kk, V,
                  o.lp_target.data_ptr(), o.accept.data_ptr(), o.n_acc.data_ptr(), o.accept_bits.data_ptr(), ws.buf.data_ptr(), ws.bytes,
                  feat.data_ptr(), 64, 5, packed.data_ptr(), 64, 32, 1, 100, 1.0, 1.0, ph.data_ptr(), Cc.data_ptr(), 1.0, N_STAGES, 0, 0,
                  None, score.data_ptr(), ks.data_ptr(), stp.data_ptr(), None, None, None, st)

    for name, (kind, fn) in fns.items():                      # settle clocks, warm every variant
        for j in range(400):
            assert launch(kind, fn, j) == 0
    torch.cuda.synchronize()
    res = {k: [] for k in fns}
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for r in range(a.rounds):
        for name, (kind, fn) in fns.items():
            for j in range(50):
                launch(kind, fn, j)
            e0.record()
            for j in range(a.reps):
                launch(kind, fn, j)
            e1.record()
            torch.cuda.synchronize()
            res[name].append(e0.elapsed_time(e1) * 1000.0 / a.reps)
    # the variants that claim the product's results must reproduce them (n_acc, accept, lp_t, score, k*), fused and plain
    # (variant 2 never consumes its siblings' slots: start from a clean workspace)
    torch.cuda.synchronize()
    ws.reset()
    ref = {}
    checks = {}
    for name, (kind, fn) in fns.items():
        n = int(name.split("/")[1]) if "/" in name else (0 if "@" in name else variants[0])
        o = bufs[0]["out"]
        for t in (o.lp_target, o.accept, o.n_acc, score, ks):
            t.zero_()
        assert launch(kind, fn, 0) == 0
        torch.cuda.synchronize()
        got = {"n_acc": o.n_acc.cpu().numpy().copy(), "accept": o.accept.cpu().numpy().copy(), "lp_t": o.lp_target.cpu().numpy().copy()}
        if kind == "fused256":
            continue
        if kind == "fused":
            got["score"] = score.cpu().numpy().copy()
            got["k_star"] = ks.cpu().numpy().copy()
        if not ref:
            ref = got
        if n in (0, 3, 4) or n >= 20:
            checks[name] = all(np.array_equal(ref[k], got[k], equal_nan=True) for k in got if k in ref)
            if not checks[name]:
                for k in got:
                    if k in ref and not np.array_equal(ref[k], got[k], equal_nan=True):
                        bad = np.argwhere(ref[k] != got[k])
                        print(f"  {name}: {k} differs at {bad[:8].tolist()}: ref {ref[k][tuple(bad[0])]} got {got[k][tuple(bad[0])]}")
    print("results equal to the product's:", checks)
    assert all(checks.values()), checks
    out = {"workload": a.workload, "rounds": a.rounds, "reps": a.reps, "results_equal": checks,
           "us_per_launch": {k: {"median": float(np.median(v)), "min": float(np.min(v)), "runs": [round(x, 3) for x in v]} for k, v in res.items()}}
    base = out["us_per_launch"]["plain"]["median"]
    for k, v in out["us_per_launch"].items():
        v["over_plain_us"] = round(v["median"] - base, 3)
        print(f"{k:10s} median {v['median']:7.3f} us  min {v['min']:7.3f}  (+{v['over_plain_us']:.3f} over plain)")
    with open(a.out, "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
