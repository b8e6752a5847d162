"""asd_draft_sample (X1: the draft tier's per-step token proposal -- temperature, nucleus / top-p truncation,
inverse-CDF draw, log q(tok)) and asd_residual_sample_ex (the residual against a nucleus-truncated draft row)
against the f64 oracle.

The reference delegates this step to HF `model.generate(do_sample=True, temperature=0.7, top_p=0.9)`
(generate_training_data.py:110-119; transformers is third party for it).  The DISTRIBUTION is pinned to HF's own
TemperatureLogitsWarper + TopPLogitsWarper (tests/golden/top_p_nucleus.npz, 54 rows; the oracle is checked against the
same fixture on the CPU); the DRAW (torch.multinomial there, inverse CDF in vocabulary order here) is RNG-specific:
parity unpinned for the token id, checked against the oracle's own definition (oracle/asd_oracle.c: oracle_draft_sample).
Bars: the nucleus threshold is bit-exact where top_p sits >= 1e-5 of mass away from the two cumulative masses that
bracket it; the token is bit-exact where, additionally, the draw is >= 1e-5 of mass away from a CDF edge; lp within
1e-5 (BASELINE: fp32 scores within 1e-5)."""
import numpy as np
import pytest

from oracle import oracle as O
from tests.helpers import encode_logits, to_device_logits

pytestmark = pytest.mark.gpu

LP_ATOL = 1e-5


@pytest.fixture(scope="module")
def K_():
    from asd_amd import kernels
    return kernels


def _rows(B, V, dtype, seed, scale=3.0):
    rng = np.random.default_rng(seed)
    x = (rng.standard_normal((B, V)) * scale).astype(np.float32)
    return encode_logits(x, dtype), rng.uniform(0, 1, B).astype(np.float32)


def _gpu(K_, store, r, B, V, dtype, inv_t, top_p):
    import torch
    lg = to_device_logits(store, dtype).view(B, V)
    samp = K_.DraftSampler(B, V, lg.dtype)
    d = samp(lg, torch.from_numpy(r).cuda(), inv_t, top_p)
    torch.cuda.synchronize()
    return d.tok.cpu().numpy(), d.lp.cpu().numpy(), d.thr.cpu().numpy()


@pytest.mark.parametrize("dtype", [O.DT_BF16, O.DT_F32, O.DT_F16])
@pytest.mark.parametrize("B,V,top_p,T", [
    (32, 152064, 0.9, 0.7),     # the reference's sampling parameters on the Qwen vocabulary
    (32, 152064, 1.0, 1.0),     # no truncation
    (8, 1000, 0.9, 0.7),        # BASELINE configs[0] vocabulary
    (5, 8, 0.5, 1.0),
    (300, 4096, 0.95, 1.3),
    (64, 32000, 0.3, 0.5),      # a tight nucleus (one to a few tokens)
])
def test_draft_sample_matches_oracle(K_, dtype, B, V, top_p, T):
    store, r = _rows(B, V, dtype, seed=B * 7 + V)
    inv_t = float(np.float32(1.0 / T))
    ref = O.draft_sample(store, dtype, r, B, V, inv_t, top_p)
    tok, lp, thr = _gpu(K_, store, r, B, V, dtype, inv_t, top_p)
    ok_p = ref["margin_p"] > 1e-5
    assert ok_p.mean() > 0.6
    assert np.array_equal(thr[ok_p], ref["thr"][ok_p])                      # the nucleus itself
    ok = ok_p & (ref["margin_r"] > 1e-5)
    assert ok.mean() > 0.4
    assert np.array_equal(tok[ok], ref["tok"][ok])
    np.testing.assert_allclose(lp[ok], ref["lp"][ok], rtol=1e-6, atol=LP_ATOL)
    # everywhere: the token is inside the nucleus the kernel reported, and lp <= 0
    x = O.logits_as_f32(store, dtype)
    assert (x[np.arange(B), tok] >= thr).all() and (lp <= 1e-6).all()
    if not (0.0 < top_p < 1.0):
        assert np.isneginf(thr).all()


@pytest.mark.parametrize("dtype", [O.DT_BF16, O.DT_F32])
def test_draft_sample_flat_and_peaked_rows_in_one_batch(K_, dtype):
    """The kernel lists a row's candidates (tokens above the (1 - top_p) / V mass floor) in LDS when they fit and sweeps the
    row when they do not: one batch holds rows of both kinds (flat rows: ~every token is a candidate) and rows in between."""
    B, V = 8, 152064
    rng = np.random.default_rng(77)
    scales = np.array([0.02, 3.0, 0.5, 6.0, 1.0, 0.1, 1.5, 2.0], np.float32)
    x = rng.standard_normal((B, V)).astype(np.float32) * scales[:, None]
    r = rng.uniform(0, 1, B).astype(np.float32)
    store = encode_logits(x, dtype)
    inv_t = float(np.float32(1.0 / 0.7))
    ref = O.draft_sample(store, dtype, r, B, V, inv_t, 0.9)
    tok, lp, thr = _gpu(K_, store, r, B, V, dtype, inv_t, 0.9)
    ok_p = ref["margin_p"] > 1e-5
    assert ok_p.sum() >= 2
    assert np.array_equal(thr[ok_p], ref["thr"][ok_p])
    ok = ok_p & (ref["margin_r"] > 1e-5)
    assert np.array_equal(tok[ok], ref["tok"][ok])
    np.testing.assert_allclose(lp[ok], ref["lp"][ok], rtol=1e-6, atol=LP_ATOL)
    xs = O.logits_as_f32(store, dtype)
    assert (xs[np.arange(B), tok] >= thr).all()
    # a flat f32 row has no token of mass > 1e-5, so top_p is never 1e-5 away from a cumulative-mass step and the exact
    # threshold is not pinned; what holds for EVERY row: the reported nucleus is the smallest one reaching top_p, to 1e-5
    for b in range(B):
        z = xs[b].astype(np.float64) * inv_t
        pr = np.exp(z - z.max())
        pr /= pr.sum()
        assert pr[xs[b] >= thr[b]].sum() >= 0.9 - 1e-5 and pr[xs[b] > thr[b]].sum() < 0.9 + 1e-5
        lp_ref = np.log(pr[tok[b]] / pr[xs[b] >= thr[b]].sum())
        assert abs(lp[b] - lp_ref) < 2e-5
    tok2, lp2, thr2 = _gpu(K_, store, r, B, V, dtype, inv_t, 0.9)
    assert np.array_equal(tok, tok2) and np.array_equal(lp, lp2) and np.array_equal(thr, thr2)


def test_draft_sample_reproduces_the_hf_warpers_nucleus(K_, golden):
    """The kernel's nucleus against transformers' TemperatureLogitsWarper + TopPLogitsWarper (the proposal distribution of
    the reference's generate() call, generate_training_data.py:110-119) on the 54 rows of tests/golden/top_p_nucleus.npz;
    log q(token) against log softmax of the warped scores."""
    from helpers import check_nucleus_against_warper, nucleus_cases
    g = golden.npz("top_p_nucleus.npz")
    n = 0
    for c in nucleus_cases(g):
        inv_t = float(np.float32(1.0) / np.float32(c["T"]))
        for r0 in (0.37, 0.93):
            tok, lp, thr = _gpu(K_, c["store"], np.array([r0], np.float32), 1, c["V"], c["dtype"], inv_t, c["top_p"])
            check_nucleus_against_warper(c, int(tok[0]), float(lp[0]), thr[0], 2e-5)
        n += 1
    assert n == 54


def test_draft_and_residual_samplers_in_a_hipgraph(K_):
    """One launch, no workspace initialisation, nothing a capture forbids: the proposal and the commit draw replay from a
    hipGraph with new uniforms in the same buffers."""
    import torch
    B, K, V = 16, 4, 32000
    store, r = _rows(B, V, O.DT_BF16, seed=91)
    lg = to_device_logits(store, O.DT_BF16).view(B, V)
    rd = torch.from_numpy(r).cuda()
    inv_t = float(np.float32(1.0 / 0.7))
    samp = K_.DraftSampler(B, V, lg.dtype)
    d = samp(lg, rd, inv_t, 0.9)                        # eager warm-up
    rng = np.random.default_rng(5)
    xt = (rng.standard_normal((B * K, V)) * 3).astype(np.float32)
    xd = (xt + rng.standard_normal((B * K, V))).astype(np.float32)
    st, sd = encode_logits(xt, O.DT_BF16), encode_logits(xd, O.DT_BF16)
    t, dd = to_device_logits(st, O.DT_BF16).view(B, K, V), to_device_logits(sd, O.DT_BF16).view(B, K, V)
    n_acc = torch.from_numpy(rng.integers(0, K, B).astype(np.int32)).cuda()
    rs = K_.ResidualSampler(B, V, t.dtype)
    tok_res = rs(t, dd, n_acc, rd, None, 1.0)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        samp(lg, rd, inv_t, 0.9, d)
        tok_res2 = rs(t, dd, n_acc, rd, None, 1.0)
    for it in range(3):
        r2 = np.random.default_rng(100 + it).uniform(0, 1, B).astype(np.float32)
        rd.copy_(torch.from_numpy(r2))
        g.replay()
        torch.cuda.synchronize()
        ref = O.draft_sample(store, O.DT_BF16, r2, B, V, inv_t, 0.9)
        ok = (ref["margin_p"] > 1e-5) & (ref["margin_r"] > 1e-5)
        assert ok.sum() >= B // 2
        assert np.array_equal(d.tok.cpu().numpy()[ok], ref["tok"][ok])
        want, margin = O.residual_sample(st, sd, O.DT_BF16, n_acc.cpu().numpy(), r2, B, K, V)
        okr = margin > 1e-5
        assert np.array_equal(tok_res2.cpu().numpy()[okr], want[okr])


def test_draft_sample_strided_rows_ties_and_masked_logits(K_):
    import torch
    B, V = 6, 4096
    rng = np.random.default_rng(3)
    x = (rng.standard_normal((B, V)) * 2).astype(np.float32)
    x[0, :] = 1.5                      # a fully tied row: the nucleus is everything, the draw uniform
    x[1, 100:] = -np.inf               # -inf logits (top-p / top-k masked rows) carry no mass
    x[2, :] = np.round(x[2, :])        # many ties at the boundary value: all of them stay inside
    x[3, 7] = 40.0                     # one token owns all the mass: nucleus of one
    r = rng.uniform(0, 1, B).astype(np.float32)
    r[4] = 0.0
    store = encode_logits(x, O.DT_F32)
    ref = O.draft_sample(store, O.DT_F32, r, B, V, 1.0, 0.9)
    pad = torch.full((B, V + 64), 1.0e4, dtype=torch.float32, device="cuda")   # poisoned padding between rows
    pad[:, :V] = torch.from_numpy(store).cuda()
    samp = K_.DraftSampler(B, V, torch.float32)
    d = samp(pad[:, :V], torch.from_numpy(r).cuda(), 1.0, 0.9)
    torch.cuda.synchronize()
    tok, lp, thr = d.tok.cpu().numpy(), d.lp.cpu().numpy(), d.thr.cpu().numpy()
    ok = (ref["margin_p"] > 1e-5) & (ref["margin_r"] > 1e-5)
    assert ok[[0, 1, 3]].all()
    assert np.array_equal(thr[ref["margin_p"] > 1e-5], ref["thr"][ref["margin_p"] > 1e-5])
    assert np.array_equal(tok[ok], ref["tok"][ok])
    np.testing.assert_allclose(lp[ok], ref["lp"][ok], atol=LP_ATOL, rtol=1e-6)
    assert thr[0] == 1.5 and abs(lp[0] - np.log(1.0 / V)) < 1e-5
    assert tok[1] < 100 and tok[3] == 7 and abs(lp[3]) < 1e-6
    # bitwise reproducible although the mass histogram is built with atomics (integer adds)
    d2 = samp(pad[:, :V], torch.from_numpy(r).cuda(), 1.0, 0.9)
    torch.cuda.synchronize()
    assert torch.equal(d2.tok, d.tok) and torch.equal(d2.lp, d.lp) and torch.equal(d2.thr, d.thr)


def test_draft_lp_feeds_the_verify_step_consistently(K_):
    """log q(tok) from the draft sampler and log p(tok) from asd_verify_accept agree when both see the same row at
    the same temperature without truncation (the acceptance ratio of a self-verified draft is 1)."""
    import torch
    B, V = 32, 152064
    store, r = _rows(B, V, O.DT_BF16, seed=11)
    lg = to_device_logits(store, O.DT_BF16).view(B, V)
    inv_t = float(np.float32(1 / 0.7))
    d = K_.DraftSampler(B, V, lg.dtype)(lg, torch.from_numpy(r).cuda(), inv_t, 1.0)
    ws = K_.VerifyWorkspace(B, 1, V, lg.dtype)
    v = K_.verify_accept(lg.view(B, 1, V), d.tok.view(B, 1), d.lp.view(B, 1).contiguous(),
                         torch.full((B, 1), 0.999, device="cuda"), ws, inv_temperature=inv_t)
    torch.cuda.synchronize()
    assert (v.lp_target.view(-1) - d.lp).abs().max().item() < 2e-6
    assert int(v.n_acc.sum()) >= B - 1


@pytest.mark.parametrize("dtype", [O.DT_BF16, O.DT_F32])
@pytest.mark.parametrize("B", [16, 112])     # 112: the one-workgroup-per-sequence form of the residual draw (B >= 96)
def test_residual_sample_against_a_nucleus_truncated_draft(K_, dtype, B):
    """asd_residual_sample_ex: p_d is the top-p truncated, renormalised draft distribution (what the drafted token
    was actually drawn from); thresholds come from asd_draft_sample on the same rows."""
    import torch
    K, V = 4, 32000 if B == 16 else 8000
    rng = np.random.default_rng(5)
    xt = (rng.standard_normal((B * K, V)) * 3).astype(np.float32)
    xd = (xt + rng.standard_normal((B * K, V))).astype(np.float32)
    st, sd = encode_logits(xt, dtype), encode_logits(xd, dtype)
    sb = encode_logits((rng.standard_normal((B, V)) * 3).astype(np.float32), dtype)
    n_acc = rng.integers(0, K + 1, B).astype(np.int32)
    r = rng.uniform(0, 1, B).astype(np.float32)
    inv_t = float(np.float32(1 / 0.7))
    dref = O.draft_sample(sd, dtype, rng.uniform(0, 1, B * K).astype(np.float32), B * K, V, inv_t, 0.9)
    thr = dref["thr"].reshape(B, K)
    want, margin = O.residual_sample(st, sd, dtype, n_acc, r, B, K, V, bonus=sb, inv_temperature=inv_t, d_threshold=thr)
    plain, _ = O.residual_sample(st, sd, dtype, n_acc, r, B, K, V, bonus=sb, inv_temperature=inv_t)
    assert (want != plain).any(), "the truncation must matter in this case"
    t = to_device_logits(st, dtype).view(B, K, V)
    d = to_device_logits(sd, dtype).view(B, K, V)
    bo = to_device_logits(sb, dtype).view(B, V)
    got = K_.ResidualSampler(B, V, t.dtype)(t, d, torch.from_numpy(n_acc).cuda(), torch.from_numpy(r).cuda(), bo, inv_t,
                                            d_threshold=torch.from_numpy(thr).cuda())
    torch.cuda.synchronize()
    ok = margin > 1e-5
    assert ok.mean() > 0.5
    assert np.array_equal(got.cpu().numpy()[ok], want[ok])


def test_draft_sample_status_codes(K_):
    import torch
    samp = K_.DraftSampler(2, 1001, torch.bfloat16)
    with pytest.raises(K_.B.AsdError):          # rows are not whole 16-byte vectors
        samp(torch.zeros((2, 1001), dtype=torch.bfloat16, device="cuda"), torch.zeros(2, device="cuda"))
    with pytest.raises(ValueError):
        K_.DraftSampler(2, 1000, torch.bfloat16)(torch.zeros((2, 1000), dtype=torch.float32, device="cuda"),
                                                  torch.zeros(2, device="cuda"))


# ---------------------------------------------------------------------------------------------------------------------
# round 3: a row spread over G workgroups inside one launch (k_draft_group)
def _run_with_groups(K_, g, store, r, B, V, dtype, inv_t, top_p):
    import torch
    with K_.test_hooks() as lib:                # the TEST build of the library: the product one has no asd_debug_* switches
        try:
            lib.asd_debug_draft_groups(int(g))
            lg = to_device_logits(store, dtype).view(B, V)
            samp = K_.DraftSampler(B, V, lg.dtype)
            d = samp(lg, torch.from_numpy(r).cuda(), inv_t, top_p)
            torch.cuda.synchronize()
            assert int(samp.buf.count_nonzero()) == 0, "every mailbox word must be handed back empty (and the status word clean)"
            return d.tok.cpu().numpy(), d.lp.cpu().numpy(), d.thr.cpu().numpy()
        finally:
            lib.asd_debug_draft_groups(0)


@pytest.mark.parametrize("dtype", [O.DT_BF16, O.DT_F16, O.DT_F32])
@pytest.mark.parametrize("B,V,top_p,T", [(8, 152064, 0.9, 0.7), (32, 152064, 0.9, 0.7), (32, 152064, 1.0, 1.0), (5, 40000, 0.5, 1.0)])
def test_results_do_not_depend_on_the_workgroups_per_row(K_, dtype, B, V, top_p, T):
    """The per-tile (max, sum) pairs are canonical and folded in a fixed order, the nucleus select is integer arithmetic:
    log q(tok), the threshold and the token must come out the same whether a row runs on one streaming workgroup
    (k_draft_row, forced with -1) or is spread over 2 ... 32 workgroups (k_draft_group) -- i.e. whatever the batch is."""
    store, r = _rows(B, V, dtype, seed=B * 7 + V)
    inv_t = float(np.float32(1.0 / T))
    base = _run_with_groups(K_, -1, store, r, B, V, dtype, inv_t, top_p)
    ref = O.draft_sample(store, dtype, r, B, V, inv_t, top_p)
    for g in (2, 4, 8, 16, 32):
        if B * g > 256:
            continue
        tok, lp, thr = _run_with_groups(K_, g, store, r, B, V, dtype, inv_t, top_p)
        assert np.array_equal(thr, base[2]), g
        assert lp.tobytes() == base[1].tobytes(), g
        # the tile masses of the streaming form come from its candidate lists (another summation order): a draw within
        # float rounding of a tile boundary of the CDF may land on the neighbouring token
        far = ref["margin_r"] > 1e-5
        assert np.array_equal(tok[far], base[0][far]), g
        assert (tok != base[0]).sum() <= 1


def test_group_kernel_reproduces_the_round2_kernel(K_, golden):
    """tests/golden/draft_sample_r02_kernel.npz: what the round-2 kernel (one workgroup per row) returned on the seeded rows
    of this file, recorded on the GPU box before the rewrite (tools/capture_draft_kernel.py; a regression pin of this repo's
    own kernel).  Without truncation everything is bit-identical (the pairs were canonical already).  With top-p the round-2
    kernel took L from a per-lane online softmax: the threshold -- an integer decision -- and the token are unchanged,
    log q differs by the rounding of L (<= 4e-7)."""
    from tools.capture_draft_kernel import CASES, DTYPES, case_rows, flat_peaked_rows
    g = golden.npz("draft_sample_r02_kernel.npz")
    n = exact_lp = 0
    for B, V, top_p, T in CASES:
        for dn, dt in DTYPES.items():
            store, r = case_rows(B, V, dt)
            tok, lp, thr = _gpu(K_, store, r, B, V, dt, float(np.float32(1.0 / T)), top_p)
            name = f"rows_{B}_{V}_{top_p}_{T}_{dn}"
            ref = O.draft_sample(store, dt, r, B, V, float(np.float32(1.0 / T)), top_p)
            far = (ref["margin_r"] > 1e-5) & (ref["margin_p"] > 1e-6)
            assert np.array_equal(thr[ref["margin_p"] > 1e-6], g[name + "/thr"][ref["margin_p"] > 1e-6]), name
            assert np.array_equal(tok[far], g[name + "/tok"][far]), name
            np.testing.assert_allclose(lp[far], g[name + "/lp"][far], rtol=0, atol=1e-6, err_msg=name)
            if not (0.0 < top_p < 1.0):
                assert lp.tobytes() == g[name + "/lp"].tobytes() and np.array_equal(tok, g[name + "/tok"]), name
            exact_lp += int((lp == g[name + "/lp"]).sum())
            n += B
    for dn in ("bf16", "f32"):
        store, r = flat_peaked_rows(DTYPES[dn])
        tok, lp, thr = _gpu(K_, store, r, 8, 152064, DTYPES[dn], float(np.float32(1.0 / 0.7)), 0.9)
        ref = O.draft_sample(store, DTYPES[dn], r, 8, 152064, float(np.float32(1.0 / 0.7)), 0.9)
        ok_p = ref["margin_p"] > 1e-5
        assert np.array_equal(thr[ok_p], g[f"flatpeaked_{dn}/thr"][ok_p])
        far = ok_p & (ref["margin_r"] > 1e-5)
        assert np.array_equal(tok[far], g[f"flatpeaked_{dn}/tok"][far])
    print(f"lp bit-identical to the round-2 kernel on {exact_lp} of {n} rows")
    assert exact_lp > n // 2


def test_group_kernel_masked_tied_and_dominant_rows(K_):
    """The edge rows of test_draft_sample_strided_rows_ties_and_masked_logits, wide enough to be spread over workgroups:
    -inf logits, a fully tied row, ties at the boundary value, one token with all the mass, r = 0, a row of -inf only."""
    import torch
    B, V = 7, 65536
    rng = np.random.default_rng(3)
    x = (rng.standard_normal((B, V)) * 2).astype(np.float32)
    x[0, :] = 1.5
    x[1, 100:] = -np.inf
    x[2, :] = np.round(x[2, :])
    x[3, 60007] = 40.0
    x[5, :] = -np.inf
    x[6, :40000] = -np.inf                                  # whole workgroups of the row see nothing but -inf
    r = rng.uniform(0, 1, B).astype(np.float32)
    r[4] = 0.0
    for dtype in (O.DT_F32, O.DT_BF16):
        store = encode_logits(x, dtype)
        ref = O.draft_sample(store, dtype, r, B, V, 1.0, 0.9)
        for g in (0, 4, 32):
            tok, lp, thr = _run_with_groups(K_, g, store, r, B, V, dtype, 1.0, 0.9)
            okp = ref["margin_p"] > 1e-5
            ok = okp & (ref["margin_r"] > 1e-5)
            assert ok[[1, 3]].all()          # (row 0: 65536 equal masses, top_p is never 1e-5 from a step; checked below)
            assert np.array_equal(thr[okp], ref["thr"][okp])
            assert np.array_equal(tok[ok], ref["tok"][ok])
            np.testing.assert_allclose(lp[ok], ref["lp"][ok], atol=LP_ATOL, rtol=1e-6)
            assert thr[0] == 1.5 and abs(lp[0] - np.log(1.0 / V)) < 1e-5
            assert tok[1] < 100 and tok[3] == 60007 and abs(lp[3]) < 1e-6
            assert tok[5] == -1 and np.isneginf(lp[5]) and tok[6] >= 40000


def test_group_kernel_in_a_hipgraph_and_across_batch_sizes(K_):
    """One workspace, zeroed once, serves calls of different batch sizes in stream order and replays from a hipGraph."""
    import torch
    V = 152064
    store, r = _rows(32, V, O.DT_BF16, seed=91)
    inv_t = float(np.float32(1.0 / 0.7))
    lg = to_device_logits(store, O.DT_BF16).view(32, V)
    rd = torch.from_numpy(r).cuda()
    samp = K_.DraftSampler(32, V, lg.dtype)
    full = samp(lg, rd, inv_t, 0.9)
    part = samp(lg[:8], rd[:8].contiguous(), inv_t, 0.9)      # G = 32 on the workspace a G = 8 call just used
    again = samp(lg, rd, inv_t, 0.9)
    torch.cuda.synchronize()
    assert torch.equal(part.tok, full.tok[:8]) and torch.equal(part.lp, full.lp[:8]) and torch.equal(part.thr, full.thr[:8])
    assert torch.equal(again.tok, full.tok) and torch.equal(again.lp, full.lp)
    assert int(samp.buf.count_nonzero()) == 0
    out = K_.DraftDraw(torch.empty_like(full.tok), torch.empty_like(full.lp), torch.empty_like(full.thr))
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(3):
            samp(lg, rd, inv_t, 0.9, out)
    for it in range(3):
        r2 = np.random.default_rng(100 + it).uniform(0, 1, 32).astype(np.float32)
        rd.copy_(torch.from_numpy(r2))
        g.replay()
        torch.cuda.synchronize()
        ref = O.draft_sample(store, O.DT_BF16, r2, 32, V, inv_t, 0.9)
        ok = (ref["margin_p"] > 1e-5) & (ref["margin_r"] > 1e-5)
        assert np.array_equal(out.tok.cpu().numpy()[ok], ref["tok"][ok])
    assert int(samp.buf.count_nonzero()) == 0


def test_a_lost_hand_off_poisons_the_row_raises_the_status_word_and_the_sampler_recovers(K_):
    """a timed-out mailbox hand-off must not only poison its row (tok = -1, lp = NaN) -- the workspace is then no
    longer all-zero (the reader gave up: a word published late is never handed back empty), so the loss is REPORTED through the
    workspace's sticky status word, the wrapper raises and re-initialises, and the next call on the same sampler is clean.
    (asd_debug_draft_withhold exists in the TEST build of the library only.)"""
    import torch
    B, V = 8, 152064
    rng = np.random.default_rng(3)
    x = (rng.standard_normal((B, V)) * 3).astype(np.float32)
    store = encode_logits(x, O.DT_BF16)
    r = rng.uniform(0, 1, B).astype(np.float32)
    ref = O.draft_sample(store, O.DT_BF16, r, B, V, 1.0, 0.9)
    lg = to_device_logits(store, O.DT_BF16).view(B, V)
    rt = torch.from_numpy(r).cuda()
    with K_.test_hooks() as lib:
        samp = K_.DraftSampler(B, V, lg.dtype)
        try:
            lib.asd_debug_draft_groups(4)
            lib.asd_debug_draft_withhold(5 * 4 + 2)                  # row 5, partner workgroup 2
            d = samp(lg, rt, 1.0, 0.9)
            torch.cuda.synchronize()
        finally:
            lib.asd_debug_draft_withhold(-1)
        tok, lp = d.tok.cpu().numpy(), d.lp.cpu().numpy()
        assert tok[5] == -1 and np.isnan(lp[5])                      # poisoned, not guessed
        ok = np.arange(B) != 5
        assert np.array_equal(tok[ok], ref["tok"][ok])               # the other rows are untouched
        assert samp.status() == K_.B.WS_LOST_HANDOFF
        with pytest.raises(K_.LostHandoffError):
            samp.check()
        assert samp.status() == 0 and int(samp.buf.count_nonzero()) == 0
        d = samp(lg, rt, 1.0, 0.9)                                   # the same sampler, the same workspace: clean again
        torch.cuda.synchronize()
        lib.asd_debug_draft_groups(0)
        assert np.array_equal(d.tok.cpu().numpy(), ref["tok"]) and samp.status() == 0


def test_hipops_check_status_raises_once_for_a_lost_hand_off_and_reinitialises(K_):
    """The loop drivers call HipOps.check_status() once per step: one synchronising read of every hand-off workspace of the
    calling thread; a raised status word becomes LostHandoffError and the workspace is re-initialised."""
    import torch
    from asd_amd.distributed import HipOps
    ops = HipOps()
    B, V = 4, 32000
    lg = (torch.randn((B, V), device="cuda") * 3).to(torch.bfloat16)
    ops.draft_sample(lg, torch.rand((B,), device="cuda"), 1.0, 0.9)
    ops.check_status()                                               # clean
    samp = next(w for w in ops._ws.values() if isinstance(w, ops.K.DraftSampler))
    samp.buf[:4].view(torch.int32).fill_(1)                          # what a kernel's timed-out wait does (ASD_WS_LOST_HANDOFF)
    with pytest.raises(ops.K.LostHandoffError):
        ops.check_status()
    ops.check_status()                                               # re-initialised
    assert int(samp.buf.count_nonzero()) == 0
