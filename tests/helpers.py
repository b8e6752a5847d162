"""Shared input builders for the parity tests (SURVEY.md §8d / BASELINE.md §3 recipe)."""
from __future__ import annotations

import numpy as np

from oracle import oracle as O

DT_NAME = {O.DT_F32: "f32", O.DT_BF16: "bf16", O.DT_F16: "f16"}


def encode_logits(x_f32: np.ndarray, dtype: int) -> np.ndarray:
    """f32 values -> the storage the kernels read (f32 array, or uint16 bit patterns)."""
    if dtype == O.DT_F32:
        return np.ascontiguousarray(x_f32, dtype=np.float32)
    if dtype == O.DT_BF16:
        return O.f32_to_bf16_bits(x_f32)
    return np.ascontiguousarray(x_f32, dtype=np.float32).astype(np.float16).view(np.uint16)


def make_verify_case(B, K, V, dtype=O.DT_BF16, seed=1234, scale=4.0, ld_row=None, margin=1e-4, n_threads=8):
    """Synthetic verify inputs + oracle outputs.

    logits ~ scale*N(0,1); tok = target arg-max w.p. 0.7 else uniform; lp_d = lp_t + N(0,0.5)
    clipped <= 0; u ~ U(0,1), re-drawn until |log u - (lp_t - lp_d)| >= margin (bit-parity set).
    Returns dict with storage array `logits` [B*K, ld_row], tok, lp_d, u and the oracle result."""
    rng = np.random.default_rng(seed)
    ld = V if ld_row is None else ld_row
    x = (rng.standard_normal((B * K, V), dtype=np.float32) * np.float32(scale))
    store = encode_logits(x, dtype)
    del x
    if ld != V:
        pad = np.zeros((B * K, ld), dtype=store.dtype)
        pad[:, :V] = store
        # poison the padding: the kernel must never read it as vocabulary
        pad[:, V:] = encode_logits(np.full((1, ld - V), 1.0e4, np.float32), dtype)
        store = pad
    xf = O.logits_as_f32(store[:, :V], dtype)
    amax = xf.argmax(axis=1).astype(np.int32)
    del xf
    tok = np.where(rng.uniform(size=B * K) < 0.7, amax, rng.integers(0, V, B * K)).astype(np.int32)
    zeros = np.zeros(B * K, np.float32)
    half = np.full(B * K, 0.5, np.float32)
    base = O.verify_accept(store, dtype, tok, zeros, half, B, K, V, ld_row=ld, n_threads=n_threads)
    lp_t = base["lp_t64"].reshape(-1)
    lp_d = np.minimum(lp_t + rng.normal(0, 0.5, B * K), 0.0).astype(np.float32)
    u = rng.uniform(0, 1, B * K).astype(np.float32)
    for _ in range(100):
        with np.errstate(divide="ignore"):
            m = np.abs(np.log(u.astype(np.float64)) - (lp_t - lp_d.astype(np.float64)))
        bad = ~(m >= margin)
        if not bad.any():
            break
        u[bad] = rng.uniform(0, 1, int(bad.sum())).astype(np.float32)
    ref = O.verify_accept(store, dtype, tok, lp_d, u, B, K, V, ld_row=ld, n_threads=n_threads)
    return dict(B=B, K=K, V=V, dtype=dtype, ld=ld, logits=store, tok=tok.reshape(B, K), lp_d=lp_d.reshape(B, K),
                u=u.reshape(B, K), ref=ref)


def to_device_logits(store: np.ndarray, dtype: int, device="cuda"):
    """Storage array -> torch tensor of the matching dtype on the GPU (no value conversion)."""
    import torch

    if dtype == O.DT_F32:
        return torch.from_numpy(store).to(device)
    t = torch.from_numpy(store.view(np.int16)).to(device)
    return t.view(torch.bfloat16 if dtype == O.DT_BF16 else torch.float16)


def run_gpu_verify(case, ws=None, **geom):
    import torch

    from asd_amd import kernels as K

    lg = to_device_logits(case["logits"], case["dtype"])
    B, Kk, V = case["B"], case["K"], case["V"]
    lg3 = lg.as_strided((B, Kk, V), (Kk * case["ld"], case["ld"], 1)) if B * Kk else lg.reshape(B, Kk, V)
    if ws is None:
        ws = K.VerifyWorkspace(max(B, 1), max(Kk, 1), V, lg.dtype)
    tok = torch.from_numpy(case["tok"]).cuda()
    lp_d = torch.from_numpy(case["lp_d"]).cuda()
    u = torch.from_numpy(case["u"]).cuda()
    r = K.verify_accept(lg3, tok, lp_d, u, ws, **geom)
    torch.cuda.synchronize()
    return dict(lp_t=r.lp_target.cpu().numpy(), accept=r.accept.cpu().numpy(), n_acc=r.n_acc.cpu().numpy(),
                bits=r.accept_bits.cpu().numpy().view(np.uint64))


# parity bar for log-probs (BASELINE.json: 1e-5 fp32); rtol covers |lp| > 10
LP_ATOL = 1e-5
LP_RTOL = 1e-6


def assert_verify_matches(got, ref, check_mask=True):
    a, b = got["lp_t"].astype(np.float64), ref["lp_t64"]
    fin = np.isfinite(b)
    np.testing.assert_allclose(a[fin], b[fin], rtol=LP_RTOL, atol=LP_ATOL)
    assert np.array_equal(np.isnan(a), np.isnan(b))
    assert np.array_equal(a[~fin & ~np.isnan(b)], b[~fin & ~np.isnan(b)])  # +-inf exactly
    if check_mask:
        assert np.array_equal(got["accept"], ref["accept"])
        assert np.array_equal(got["n_acc"], ref["n_acc"])
        assert np.array_equal(got["bits"], ref["bits"])


# The reference's per-token idiom (softmax -> index -> log, torch CPU f32) at V = 152064 is itself 0.3e-5 .. 1.7e-5 away
# from the exact (f64) log-prob: torch's f32 accumulation of 152064 exponentials.  An exact kernel therefore cannot sit
# within 1e-5 of THAT number on full-vocabulary rows; it is held to 1e-5 of the f64 oracle and to this bound of the idiom.
REF_F32_SUM_ERR = 2.5e-5


# ---- full-size reference-idiom goldens (tests/golden/logprob_idiom_full.npz, oracle/gen_golden.py) ----------------
def full_size_row(seed: int, row: int, vocab: int = 152064) -> np.ndarray:
    """The f32 score row the generator drew for (seed, row): the fixture stores seeds, not 15 MB of scores."""
    return (np.random.default_rng([seed, row]).standard_normal(vocab) * 4.0).astype(np.float32)


def full_size_cases(g):
    """Yield (variant, tok, expected logprob, x16 f32 values after 16-bit storage (or raw f32), keep ids or None)."""
    seed, V = int(g["seed"]), int(g["vocab"])
    names = [str(v) for v in g["variants"]]
    for i in range(g["tok"].shape[0]):
        var = names[int(g["variant"][i])]
        x = full_size_row(seed, int(g["row"][i]), V)
        if var.startswith("bf16"):
            x = O.bf16_bits_to_f32(O.f32_to_bf16_bits(x))
        elif var.startswith("f16"):
            x = x.astype(np.float16).astype(np.float32)
        keep = g["keep"][int(g["keep_off"][i]):int(g["keep_off"][i + 1])] if var.endswith("_topp") else None
        yield var, int(g["tok"][i]), float(g["logprob"][i]), x, keep
