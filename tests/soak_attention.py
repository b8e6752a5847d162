"""Soak of asd_rope_kv_store + asd_attn_ragged on random geometries (run by hand on a GPU box, not collected):
    python tests/soak_attention.py [cases] [seed]
Random (B, T, H, KVH, t_max), random per-sequence positions incl. 0 and a full cache, random subset rows: the cache the rope
kernel writes is fed to the attention kernel, whose output is checked against an f64 softmax over the SAME cache
(|o - ref| <= 2^-7 |ref| + 6e-3, as in tests/test_gpu_decoder.py)."""
import math
import sys
from importlib import import_module
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
K_ = import_module("adaptive-speculative-decoding_amd.kernels")
B_ = import_module("adaptive-speculative-decoding_amd._binding")
BF = torch.bfloat16


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = np.random.default_rng(seed)
    lib = K_._lib()
    for c in range(cases):
        KVH = int(rng.choice([1, 2, 4, 8]))
        rep = int(rng.choice([1, 2, 4, 5, 7, 8]))
        H = KVH * rep
        Bn = int(rng.integers(1, 9))
        T = int(rng.choice([1, 1, 2, 5, 9, 17, 40]))
        t_max = 32 * int(rng.integers(max(1, (T + 31) // 32), 13))
        Bc = Bn + int(rng.integers(0, 3))
        g = torch.Generator(device="cuda").manual_seed(int(rng.integers(0, 2 ** 31)))
        kc = torch.randn(Bc, KVH, t_max, 128, generator=g, device="cuda").to(BF)
        v = torch.randn(Bc, KVH, t_max, 128, generator=g, device="cuda").to(BF)
        vt = v.transpose(2, 3).contiguous()
        M = Bn * T
        width = (H + 2 * KVH) * 128
        qkv = torch.randn(M, width, generator=g, device="cuda").to(BF)
        pos0 = torch.from_numpy(rng.integers(0, t_max - T + 1, Bn)).cuda()
        if rng.random() < 0.5:
            pos0[0] = 0
        if rng.random() < 0.5:
            pos0[-1] = t_max - T
        pos = (pos0[:, None] + torch.arange(T, device="cuda")).reshape(M).to(torch.int32)
        rows = torch.from_numpy(rng.permutation(Bc)[:Bn].astype(np.int32)).cuda() if rng.random() < 0.5 else None
        inv = (1.0 / (1.0e6 ** (torch.arange(0, 128, 2, device="cuda", dtype=torch.float32) / 128))).contiguous()
        rp = None if rows is None else rows.data_ptr()
        B_.check("asd_rope_kv_store", lib.asd_rope_kv_store(qkv.data_ptr(), width, pos.data_ptr(), rp, inv.data_ptr(), B_.DTYPE_BF16, Bn, T, H,
                                                            KVH, 128, kc.data_ptr(), vt.data_ptr(), t_max, None))
        out = torch.empty(M, H * 128, dtype=BF, device="cuda")
        B_.check("asd_attn_ragged", lib.asd_attn_ragged(qkv.data_ptr(), width, kc.data_ptr(), vt.data_ptr(), pos.data_ptr(), rp, B_.DTYPE_BF16,
                                                        Bn, T, H, KVH, 128, t_max, out.data_ptr(), H * 128, None))
        q = qkv.double().view(M, H + 2 * KVH, 128)[:, :H]               # (rotated in place by the rope kernel)
        vv_all = vt.transpose(2, 3)
        for m in range(M):
            b = m // T
            row = int(rows[b]) if rows is not None else b
            L = int(pos[m]) + 1
            kk = kc[row, :, :L].double().repeat_interleave(rep, dim=0)
            vv = vv_all[row, :, :L].double().repeat_interleave(rep, dim=0)
            sc = torch.einsum("hd,hld->hl", q[m], kk) / math.sqrt(128.0)
            ref = torch.einsum("hl,hld->hd", torch.softmax(sc, dim=-1), vv)
            got = out[m].double().view(H, 128)
            excess = ((got - ref).abs() - (2.0 ** -7 * ref.abs() + 6e-3)).max().item()
            assert excess <= 0.0, f"case {c}: B={Bn} T={T} H={H} KVH={KVH} t_max={t_max} m={m}: excess {excess:.3e}"
        if c % 10 == 0:
            print(f"case {c}: B={Bn} T={T} H={H} KVH={KVH} t_max={t_max} rows={'subset' if rows is not None else 'all'} ok", flush=True)
    print(f"soak_attention: {cases} cases passed (seed {seed})")


if __name__ == "__main__":
    main()
