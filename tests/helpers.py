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


# ---- top-p goldens (tests/golden/top_p_nucleus.npz: HF TemperatureLogitsWarper + TopPLogitsWarper, oracle/gen_golden.py) ----
def nucleus_cases(g):
    """Yield dict(row, V, dtype, T, top_p, x (f32 values after the storage rounding), store (what the kernels read), n_keep,
    thr, ties_removed, lse_keep, margin) for every row of the fixture; rows are regenerated from (seed, row)."""
    seed = int(g["seed"])
    dts = {"f32": O.DT_F32, "bf16": O.DT_BF16, "f16": O.DT_F16}
    for i in range(g["row"].shape[0]):
        V, storage = int(g["V"][i]), str(g["storage"][i])
        x = (np.random.default_rng([seed, int(g["row"][i])]).standard_normal(V) * float(g["scale"][i])).astype(np.float32)
        store = encode_logits(x[None, :], dts[storage])
        yield dict(row=int(g["row"][i]), V=V, dtype=dts[storage], T=float(g["T"][i]), top_p=float(g["top_p"][i]),
                   x=O.logits_as_f32(store, dts[storage])[0], store=store, n_keep=int(g["n_keep"][i]), thr=np.float32(g["thr"][i]),
                   ties_removed=int(g["ties_removed"][i]), lse_keep=float(g["lse_keep"][i]), margin=float(g["margin"][i]))


def check_nucleus_against_warper(c, tok, lp, thr, lp_atol):
    """One draft-sampler result (token, log q(token), reported threshold) against the HF warper's nucleus of the row."""
    x, T = c["x"], np.float32(c["T"])
    if c["margin"] > 1e-5:                 # top_p is not within 1e-5 of a cumulative-mass step: the cut is unambiguous
        assert np.float32(thr) == c["thr"], (c["row"], thr, c["thr"])
        assert int((x >= thr).sum()) == c["n_keep"] + c["ties_removed"]       # the kernel keeps every tie, the warper's sort order decides
    else:                                   # (f32 cumsum of the warper vs the exact masses): at most one step apart
        assert abs(int((x >= thr).sum()) - c["n_keep"] - c["ties_removed"]) <= max(2, c["ties_removed"] + 2), c["row"]
    assert x[tok] >= thr
    if c["margin"] > 1e-5 and c["ties_removed"] == 0:
        want = float(x[tok]) * float(np.float32(1.0) / T) - c["lse_keep"]      # log softmax(warped scores)[tok]
        assert abs(lp - want) < lp_atol, (c["row"], lp, want)


# ---- A5 + residual goldens (tests/golden/speculative_sampling.npz: transformers' _speculative_sampling) ------------------
def spec_cases(g):
    """Yield the verify / residual inputs of every fixture case, regenerated from (seed, case) exactly as
    oracle/gen_golden.py::spec_case_inputs drew them, with HF's n_matches and the inverse-CDF tokens of its p'."""
    seed = int(g["seed"])
    for i in range(g["case"].shape[0]):
        K, V = int(g["K"][i]), int(g["V"][i])
        rng = np.random.default_rng([seed, int(g["case"][i])])
        cand = (rng.standard_normal((K, V)) * float(g["scale"][i])).astype(np.float32)
        new = np.empty((K + 1, V), np.float32)
        new[:K] = (cand.astype(np.float64) + rng.standard_normal((K, V)) * float(g["spread"][i])).astype(np.float32)
        new[K] = (rng.standard_normal(V) * float(g["scale"][i])).astype(np.float32)
        pick = rng.uniform(0, 1, K)
        ids = np.empty(K, np.int32)
        lq = np.empty(K)
        for k in range(K):
            z = cand[k].astype(np.float64)
            q = np.exp(z - z.max())
            q /= q.sum()
            ids[k] = min(int(np.searchsorted(np.cumsum(q), pick[k], side="right")), V - 1)
            lq[k] = np.log(q[ids[k]])
        u = g["u"][int(g["u_off"][i]):int(g["u_off"][i + 1])].astype(np.float32)
        yield dict(K=K, V=V, cand=cand, new=new, tok=ids, lp_d=lq.astype(np.float32), u=u, n_matches=int(g["n_matches"][i]),
                   r=g["r"][i].astype(np.float32), want_tok=g["tok"][i].astype(np.int32), margin=g["margin"][i])


# ---- A5 + residual + proposal at V = 152064 on bf16 / f16 rows with T = 0.7, top-p 0.9
# (tests/golden/speculative_sampling_full.npz: transformers' warpers + _speculative_sampling)
def spec_full_cases(g):
    """Yield, per fixture case, the RAW rows as stored (regenerated from (seed, case) exactly as
    oracle/gen_golden.py::spec_full_rows drew them) with HF's results: drafted tokens, log q(token), nucleus thresholds in
    raw score units, acceptance uniforms, n_matches, residual-draw tokens."""
    seed, V = int(g["seed"]), int(g["V"])
    dts = {"bf16": O.DT_BF16, "f16": O.DT_F16}
    for i in range(g["case"].shape[0]):
        K, storage = int(g["K"][i]), str(g["storage"][i])
        rng = np.random.default_rng([seed, int(g["case"][i])])
        cand = (rng.standard_normal((K, V)) * float(g["scale"][i])).astype(np.float32)
        new = np.empty((K + 1, V), np.float32)
        new[:K] = (cand.astype(np.float64) + rng.standard_normal((K, V)) * float(g["spread"][i])).astype(np.float32)
        new[K] = (rng.standard_normal(V) * float(g["scale"][i])).astype(np.float32)
        pick = rng.uniform(0, 1, K).astype(np.float32)
        a, b = int(g["off"][i]), int(g["off"][i + 1])
        dt = dts[storage]
        yield dict(case=int(g["case"][i]), K=K, V=V, dtype=dt, cand=encode_logits(cand, dt), new=encode_logits(new, dt), pick=pick,
                   tok=g["ids"][a:b].astype(np.int32), lq=g["lq"][a:b], thr=g["thr"][a:b].astype(np.float32),
                   n_keep=g["n_keep"][a:b], ties_removed=g["ties_removed"][a:b], pick_margin=g["pick_margin"][a:b],
                   u=g["u"][a:b].astype(np.float32),
                   n_matches=int(g["n_matches"][i]), r=g["r"][i].astype(np.float32), want_tok=g["tok"][i].astype(np.int32),
                   margin=g["margin"][i], inv_t=float(np.float32(1.0) / np.float32(g["T"])), top_p=float(g["top_p"]))
