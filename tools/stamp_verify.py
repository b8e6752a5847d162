#!/usr/bin/env python3
"""Diagnostic: where does a verify launch spend its time?  Builds a SEPARATE library with -DASD_STAMP
(gpurun_out/libasd_hip_stamp.so; the shipped library never contains stamps), runs a workload and
prints, relative to the earliest workgroup start, the distribution of
  t0 start | t1 first batch consumed | t2 stream end | t3 workgroup reduced | t4 before ticket | t5 done.
Stamp = s_memrealtime (100 MHz => 10 ns ticks).

    python tools/stamp_verify.py [c3|c2|c5] [splits,threads,unroll,nt] [chain] [fused]

`fused`: stamp asd_verify_accept_fused_ex instead (t4 = finisher starts waiting for the hand-off slots, t5 = has them,
t8 = in-kernel epilogue done).
"""
import ctypes as C
import math
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from bench import WORKLOADS, build_inputs  # noqa: E402


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
    geom = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 0, 0, -1]
    chain = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    fused = len(sys.argv) > 4 and sys.argv[4] in ("fused", "fused2")
    twice = len(sys.argv) > 4 and sys.argv[4] == "fused2"    # the in-kernel epilogue run twice: cold vs warm instruction cache
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    lib_path = os.path.join(out, "libasd_hip_stamp.so")
    csrc = os.path.join(ROOT, "adaptive-speculative-decoding_amd", "csrc")
    subprocess.check_call(["hipcc", "-std=c++17", "-O3", "--offload-arch=gfx950", "-fPIC", "-shared", "-DASD_STAMP", *( ["-DASD_EPI_TWICE"] if twice else []),
                           "-ffp-contract=off", "-mllvm", "-amdgpu-kernarg-preload-count=11", f"-I{ROOT}/include", f"-I{csrc}", os.path.join(csrc, "verify_accept.hip"),
                           os.path.join(csrc, "api.hip"), os.path.join(csrc, "predictor.hip"), "-o", lib_path])
    lib = C.CDLL(lib_path)
    from asd_amd import kernels as K
    B, Kk, V, _ = WORKLOADS[wl]
    dev = torch.device("cuda", 0)
    nbuf = max(3, math.ceil(640e6 / (B * Kk * V * 2)))
    ws, bufs = build_inputs(torch, K, B, Kk, V, nbuf, dev, 1234)
    nblk = B * Kk * 64
    stamps = torch.zeros((nblk * 32,), dtype=torch.int64, device=dev)
    lib.asd_debug_set_stamp_buffer.argtypes = [C.c_void_p]
    assert lib.asd_debug_set_stamp_buffer(stamps.data_ptr()) == 0
    from asd_amd._binding import verify_options
    fn = lib.asd_verify_accept_ex
    fn.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    opt = verify_options(1.0, *geom)
    st = torch.cuda.current_stream().cuda_stream
    if fused:
        from bench import N_STAGES, STAGE_COSTS, predictor_weights
        from asd_amd._binding import SIGNATURES
        plain = fn
        ff = lib.asd_verify_accept_fused_ex
        ff.restype, ff.argtypes = SIGNATURES["asd_verify_accept_fused_ex"]
        packed = K.pack_mlp_weights(*predictor_weights(np), device=dev)
        feat = torch.from_numpy((np.random.default_rng(7).standard_normal((B, 64)) * 0.3).astype(np.float32)).to(dev)
        Cc = torch.tensor(STAGE_COSTS, dtype=torch.float64, device=dev)
        ph = torch.ones((B, N_STAGES), dtype=torch.float64, device=dev)
        score = torch.empty((B,), dtype=torch.float32, device=dev)
        ks = torch.empty((B,), dtype=torch.int32, device=dev)
        stp = torch.empty((B,), dtype=torch.uint8, device=dev)

        def fn(lg, dt, ld, tok, lpd, u, B_, K_, V_, lp, acc, nacc, bits, wsb, wsn, optp, stream):   # noqa: E306
            return ff(lg, dt, ld, tok, lpd, u, B_, K_, V_, lp, acc, nacc, bits, wsb, wsn, feat.data_ptr(), 64, 5, packed.data_ptr(),
                      64, 32, 1, 100, 1.0, 1.0, ph.data_ptr(), Cc.data_ptr(), 1.0, N_STAGES, 0, 0, None, score.data_ptr(),
                      ks.data_ptr(), stp.data_ptr(), None, None, optp, stream)
    res, xcc, waves = [], [], []
    if chain > 1:
        for j in range(600):   # settle clocks like bench.py does
            bj = bufs[j % nbuf]
            oj = bj['out']
            fn(bj['logits'].data_ptr(), 1, V, bj['tok'].data_ptr(), bj['lp_d'].data_ptr(), bj['u'].data_ptr(), B, Kk, V, oj.lp_target.data_ptr(), oj.accept.data_ptr(), oj.n_acc.data_ptr(), oj.accept_bits.data_ptr(), ws.buf.data_ptr(), ws.bytes, C.addressof(opt), st)
    for it in range(12):
        buf = bufs[it % nbuf]
        o = buf["out"]
        stamps.zero_()
        torch.cuda.synchronize()
        # steady state: `chain` back-to-back launches, the stamps that remain are the LAST launch's
        for j in range(chain):
            bj = bufs[(it + j) % nbuf]
            oj = bj["out"]
            rc = fn(bj["logits"].data_ptr(), 1, V, bj["tok"].data_ptr(), bj["lp_d"].data_ptr(), bj["u"].data_ptr(), B, Kk,
                    V, oj.lp_target.data_ptr(), oj.accept.data_ptr(), oj.n_acc.data_ptr(), oj.accept_bits.data_ptr(),
                    ws.buf.data_ptr(), ws.bytes, C.addressof(opt), st)
            assert rc == 0, rc
        torch.cuda.synchronize()
        raw = stamps.cpu().numpy()
        cus = K.device_cu_count()
        auto_s = 1 if B * Kk >= cus else min(-(-cus // (B * Kk)), (V * 2) // (8 * 2 * 1024))
        grid = B * Kk * (geom[0] if geom[0] > 0 else auto_s)   # the launcher's heuristic: one workgroup per row at rows >= CUs
        s = raw[: grid * 16].reshape(grid, 16)
        wv = raw[grid * 16: grid * 16 + grid * 16].reshape(grid, 16).astype(np.float64)
        live = s[:, 0] > 0
        s = s[live].astype(np.float64)
        t0 = s[:, 0].min()
        rel = (s[:, [0, 1, 2, 3, 4, 5, 7, 8, 9, 10, 11, 12, 13, 14, 15]] - t0) / 100.0      # us
        if it >= 2:
            res.append(rel)
            xcc.append(s[:, 6].astype(int))
            waves.append((wv - t0) / 100.0)
    rel = np.concatenate(res)
    names = ["start", "first batch consumed", "stream end", "wg reduced", "before ticket / slot wait", "ticket back / slots read",
             "first loads issued", "in-kernel epilogue done", "epi: statistics done", "epi: both layers done", "epi: sigmoid done",
             "epi pass 1: statistics", "epi pass 1: layers", "epi pass 1: sigmoid", "epi pass 2 done"]
    print(f"workload {wl}, geometry {geom}, chain {chain}, {len(res)} launches, {rel.shape[0] // len(res)} workgroups each; us from first start")
    for i, n in enumerate(names):
        col = rel[:, i]
        col = col[col >= 0]
        if col.size == 0:
            continue
        print(f"  {n:22s} min {col.min():7.2f}  p50 {np.median(col):7.2f}  p90 {np.percentile(col, 90):7.2f}  max {col.max():7.2f}")
    per = rel[:, 2] - rel[:, 1]
    print(f"  stream phase (t2-t1)   p50 {np.median(per):7.2f}  max {per.max():7.2f}")
    print(f"  ramp (t1-t0)           p50 {np.median(rel[:, 1] - rel[:, 0]):7.2f}  max {(rel[:, 1] - rel[:, 0]).max():7.2f}")
    ok = rel[:, 5] > 0
    if ok.any():
        print(f"  tail (t5-t2)           p50 {np.median((rel[:, 5] - rel[:, 2])[ok]):7.2f}  max {(rel[:, 5] - rel[:, 2])[ok].max():7.2f}")
    fin = rel[:, 7] > 0          # finishers: the workgroups that ran the in-kernel epilogue
    if fused and fin.any():
        f = rel[fin]

        def seg(name, a, b):
            d = f[:, b] - f[:, a]
            print(f"  finisher {name:34s} p50 {np.median(d):6.2f}  p90 {np.percentile(d, 90):6.2f}  max {d.max():6.2f}")
        seg("stream end -> wg reduced", 2, 3)
        seg("wg reduced -> slot wait begins", 3, 4)
        seg("slot wait (K-1 polls)", 4, 5)
        if twice:
            seg("pass 1: slots -> statistics", 5, 11)
            seg("pass 1: statistics -> both layers", 11, 12)
            seg("pass 1: layers -> sigmoid", 12, 13)
            seg("pass 1: sigmoid -> decided+stored", 13, 7)
            seg("pass 1 total", 5, 7)
            seg("pass 2: -> statistics", 7, 8)
            seg("pass 2: statistics -> both layers", 8, 9)
            seg("pass 2: layers -> sigmoid", 9, 10)
            seg("pass 2: sigmoid -> decided+stored", 10, 14)
            seg("pass 2 total", 7, 14)
        else:
            seg("epi: slots -> statistics", 5, 8)
            seg("epi: statistics -> both layers", 8, 9)
            seg("epi: layers -> sigmoid", 9, 10)
            seg("epi: sigmoid -> decided+stored", 10, 7)
            seg("epi total", 5, 7)
        seg("stream end -> all done", 2, 14 if twice else 7)


    w = np.concatenate(waves)
    w = w[:, (w > 0).all(axis=0)]
    print("  per-wave stream end by wave index (p50 over workgroups):")
    print("    " + " ".join(f"{np.median(w[:, i]):5.2f}" for i in range(w.shape[1])))
    print(f"    slowest-wave p50 {np.median(w.max(axis=1)):5.2f}  mean-wave p50 {np.median(w.mean(axis=1)):5.2f}  fastest-wave p50 {np.median(w.min(axis=1)):5.2f}")
    return rel, np.concatenate(xcc)


def per_xcd(rel, xcc):
    print("  per-XCD stream end (t2) p50 / max, and workgroups seen:")
    for x in sorted(set(xcc.tolist())):
        col = rel[xcc == x, 2]
        print(f"    xcd {x}: p50 {np.median(col):6.2f}  max {col.max():6.2f}  n {col.size}")


if __name__ == "__main__":
    r, x = main()
    per_xcd(r, x)
