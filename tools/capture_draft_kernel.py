#!/usr/bin/env python3
"""Record what the CURRENTLY BUILT asd_draft_sample returns on the seeded rows of tests/test_gpu_draft_sample.py
(run on the GPU box):  python tools/capture_draft_kernel.py gpurun_out/draft_kernel_capture.npz

Used once at the start of round 3, on the round-2 kernel (one 1024-lane workgroup per row), to pin the rewrite that
spreads a row over several workgroups: tests/golden/draft_sample_r02_kernel.npz holds tok / lp / thr per case
(inputs are regenerated from the seeds; the file is a few KB).  A regression pin of this repo's own kernel --
not reference data (the reference delegates the proposal to HF generate(), generate_training_data.py:110-119)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import oracle as O                      # noqa: E402  (storage encoders only)
from tests.helpers import encode_logits, to_device_logits   # noqa: E402

CASES = [  # (B, V, top_p, T) of test_draft_sample_matches_oracle
    (32, 152064, 0.9, 0.7), (32, 152064, 1.0, 1.0), (8, 1000, 0.9, 0.7), (5, 8, 0.5, 1.0), (300, 4096, 0.95, 1.3),
    (64, 32000, 0.3, 0.5), (8, 152064, 0.9, 0.7), (128, 152064, 0.9, 0.7), (8, 152064, 1.0, 0.7),
]
DTYPES = {"bf16": O.DT_BF16, "f32": O.DT_F32, "f16": O.DT_F16}


def case_rows(B, V, dtype, scale=3.0):
    rng = np.random.default_rng(B * 7 + V)
    x = (rng.standard_normal((B, V)) * scale).astype(np.float32)
    return encode_logits(x, dtype), rng.uniform(0, 1, B).astype(np.float32)


def flat_peaked_rows(dtype):
    B, V = 8, 152064
    rng = np.random.default_rng(77)
    scales = np.array([0.02, 3.0, 0.5, 6.0, 1.0, 0.1, 1.5, 2.0], np.float32)
    x = rng.standard_normal((B, V)).astype(np.float32) * scales[:, None]
    r = rng.uniform(0, 1, B).astype(np.float32)
    return encode_logits(x, dtype), r


def main(out_path):
    import torch
    from asd_amd import kernels as K

    out = {}

    def run(name, store, r, B, V, dtype, inv_t, top_p):
        lg = to_device_logits(store, dtype).view(B, V)
        d = K.DraftSampler(B, V, lg.dtype)(lg, torch.from_numpy(r).cuda(), inv_t, top_p)
        torch.cuda.synchronize()
        out[name + "/tok"] = d.tok.cpu().numpy()
        out[name + "/lp"] = d.lp.cpu().numpy()
        out[name + "/thr"] = d.thr.cpu().numpy()

    for B, V, top_p, T in CASES:
        for dn, dt in DTYPES.items():
            store, r = case_rows(B, V, dt)
            run(f"rows_{B}_{V}_{top_p}_{T}_{dn}", store, r, B, V, dt, float(np.float32(1.0 / T)), top_p)
    for dn in ("bf16", "f32"):
        store, r = flat_peaked_rows(DTYPES[dn])
        run(f"flatpeaked_{dn}", store, r, 8, 152064, DTYPES[dn], float(np.float32(1.0 / 0.7)), 0.9)
    os.makedirs(os.path.dirname(os.path.abspath(out_path)), exist_ok=True)
    np.savez_compressed(out_path, **out)
    print(f"wrote {out_path}: {len(out) // 3} cases")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "draft_kernel_capture.npz"))
