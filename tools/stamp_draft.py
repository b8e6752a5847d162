#!/usr/bin/env python3
"""Diagnostic: where does an asd_draft_sample launch spend its time?  Builds a SEPARATE library with -DASD_STAMP
(gpurun_out/libasd_draft_stamp.so; the shipped library never contains stamps), runs the STREAMING form of the sampler
(k_draft_row, what B > 128 uses; the group kernel of round 3 is timed by tools/bench_sampling.py) on bench-shaped rows and
prints the phase boundaries of workgroup 0 in microseconds (s_memrealtime, 100 MHz => 10 ns ticks):
  0 start | 1 row LSE | 2,3 histogram level 1 sweep | 4,5 level 2 | 6,7 level 3 (f32 only) | 8 threshold known |
  9 nucleus LSE | 10 tile masses | 11 token written.

    python tools/stamp_draft.py [B] [top_p]
"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    top_p = float(sys.argv[2]) if len(sys.argv) > 2 else 0.9
    scale = float(sys.argv[3]) if len(sys.argv) > 3 else 3.0
    V = 152064
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    lib_path = os.path.join(out, "libasd_draft_stamp.so")
    csrc = os.path.join(ROOT, "adaptive-speculative-decoding_amd", "csrc")
    subprocess.check_call(["hipcc", "-std=c++17", "-O3", "--offload-arch=gfx950", "-fPIC", "-shared", "-DASD_STAMP", "-DASD_TEST_HOOKS",
                           f"-I{ROOT}/include", f"-I{csrc}", os.path.join(csrc, "draft_sample.hip"), os.path.join(csrc, "api.hip"), "-o", lib_path])
    lib = C.CDLL(lib_path)
    lib.asd_debug_draft_groups(-1)       # the stamps live in k_draft_row, the one-workgroup-per-row streaming form (B > 128)
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(5)
    logits = (torch.randn((B, V), generator=g, device=dev) * scale).to(torch.bfloat16)
    r = torch.rand((B,), generator=g, device=dev)
    tok = torch.zeros((B,), dtype=torch.int32, device=dev)
    lp = torch.zeros((B,), dtype=torch.float32, device=dev)
    thr = torch.zeros((B,), dtype=torch.float32, device=dev)
    ws = torch.zeros((4096,), dtype=torch.uint8, device=dev)
    fn = lib.asd_draft_sample
    fn.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_void_p,
                   C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    st = torch.cuda.current_stream().cuda_stream
    stamps = (C.c_ulonglong * 16)()
    rows = []
    for it in range(40):
        rc = fn(logits.data_ptr(), V, 1, r.data_ptr(), B, V, 1.0 / 0.7, top_p, tok.data_ptr(), lp.data_ptr(), thr.data_ptr(),
                ws.data_ptr(), 4096, st)
        assert rc == 0, rc
        torch.cuda.synchronize()
        assert lib.asd_debug_draft_stamps(stamps) == 0
        s = np.array(list(stamps), dtype=np.int64)
        rows.append((s - s[0]) / 100.0)
    med = np.median(np.array(rows[10:]), axis=0)
    print("B", B, "top_p", top_p, "phase boundaries (us, workgroup 0):")
    for i in (1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11):
        print(f"  {i:2d}: {med[i]:8.2f}")


if __name__ == "__main__":
    main()
