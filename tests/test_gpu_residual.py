"""asd_residual_sample (the token committed after the accepted prefix) against the f64 oracle.
No reference symbol exists: parity unpinned; integer result, so the bar is bit-exact
on the draws whose distance to a CDF edge exceeds 1e-5 of the total mass, and membership of the
support everywhere."""
import numpy as np
import pytest

from oracle import oracle as O
from tests.helpers import encode_logits, to_device_logits

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def K_():
    from asd_amd import kernels
    return kernels


def _case(B, K, V, dtype, seed, spread=1.0):
    rng = np.random.default_rng(seed)
    xt = (rng.standard_normal((B * K, V)) * 3).astype(np.float32)
    xd = (xt + rng.standard_normal((B * K, V)) * spread).astype(np.float32)
    bonus = (rng.standard_normal((B, V)) * 3).astype(np.float32)
    n_acc = rng.integers(0, K + 1, B).astype(np.int32)
    r = rng.uniform(0, 1, B).astype(np.float32)
    return encode_logits(xt, dtype), encode_logits(xd, dtype), encode_logits(bonus, dtype), n_acc, r


def _gpu(K_, st, sd, sb, n_acc, r, B, K, V, dtype, inv_t=1.0, with_bonus=True):
    import torch
    t = to_device_logits(st, dtype).view(B, K, V)
    d = to_device_logits(sd, dtype).view(B, K, V)
    bo = to_device_logits(sb, dtype).view(B, V) if with_bonus else None
    samp = K_.ResidualSampler(B, V, t.dtype)
    out = samp(t, d, torch.from_numpy(n_acc).cuda(), torch.from_numpy(r).cuda(), bo, inv_t)
    torch.cuda.synchronize()
    return out.cpu().numpy()


@pytest.mark.parametrize("dtype", [O.DT_BF16, O.DT_F32, O.DT_F16])
@pytest.mark.parametrize("B,K,V", [(32, 8, 152064), (5, 4, 1000), (3, 2, 8), (64, 8, 32000), (300, 3, 4096),
                                   (128, 2, 152064)])     # B >= 96: one 1024-lane workgroup per sequence (k_residual_row)
def test_residual_sample_matches_oracle(K_, dtype, B, K, V):
    st, sd, sb, n_acc, r = _case(B, K, V, dtype, seed=B + V)
    want, margin = O.residual_sample(st, sd, dtype, n_acc, r, B, K, V, bonus=sb)
    got = _gpu(K_, st, sd, sb, n_acc, r, B, K, V, dtype)
    ok = margin > 1e-5
    assert ok.mean() > 0.5            # at V = 152064 many tokens own less than 2e-5 of the mass
    assert np.array_equal(got[ok], want[ok])
    # everywhere: the token lies in the support of the distribution it was drawn from
    xt = O.logits_as_f32(st, dtype).astype(np.float64)
    xd = O.logits_as_f32(sd, dtype).astype(np.float64)
    for b in np.where(~ok)[0]:
        assert abs(int(got[b]) - int(want[b])) <= 64 * 8       # a boundary case can only move to a neighbour
        j = n_acc[b]
        if j < K:
            def sm(x):
                e = np.exp(x - x.max())
                return e / e.sum()
            assert sm(xt[b * K + j])[got[b]] > 0


def test_temperature_degenerate_and_missing_bonus(K_):
    B, K, V = 16, 4, 8192
    st, sd, sb, n_acc, r = _case(B, K, V, O.DT_BF16, seed=9)
    for inv_t in (0.5, 1.0 / 0.7):
        want, margin = O.residual_sample(st, sd, O.DT_BF16, n_acc, r, B, K, V, bonus=sb, inv_temperature=np.float32(inv_t))
        got = _gpu(K_, st, sd, sb, n_acc, r, B, K, V, O.DT_BF16, inv_t=float(np.float32(inv_t)))
        ok = margin > 1e-5
        assert np.array_equal(got[ok], want[ok])
    # identical target and draft rows: the residual is empty, the draw falls back to p_t
    n0 = np.zeros(B, np.int32)
    want, margin = O.residual_sample(st, st, O.DT_BF16, n0, r, B, K, V, bonus=sb)
    got = _gpu(K_, st, st, sb, n0, r, B, K, V, O.DT_BF16)
    assert np.array_equal(got[margin > 1e-5], want[margin > 1e-5])
    # every token accepted but no bonus logits supplied: -1
    nK = np.full(B, K, np.int32)
    assert (_gpu(K_, st, sd, sb, nK, r, B, K, V, O.DT_BF16, with_bonus=False) == -1).all()
    # r = 0 picks the first token with mass, r -> 1 the last
    r0 = np.zeros(B, np.float32)
    want, _ = O.residual_sample(st, sd, O.DT_BF16, n_acc, r0, B, K, V, bonus=sb)
    assert np.array_equal(_gpu(K_, st, sd, sb, n_acc, r0, B, K, V, O.DT_BF16), want)


def test_rows_must_be_whole_vectors(K_):
    import torch
    t = torch.zeros((2, 2, 1001), dtype=torch.bfloat16, device="cuda")
    samp = K_.ResidualSampler(2, 1001)
    with pytest.raises(K_.B.AsdError):
        samp(t, t, torch.zeros(2, dtype=torch.int32, device="cuda"), torch.zeros(2, device="cuda"))


# ---------------------------------------------------------------------------------------------------------------------
# round 3: a sequence's two rows spread over G workgroups inside one launch (k_residual_group)
def _residual_with_groups(K_, g, t, d, bo, n_acc, r, inv_t, thr=None):
    import torch
    B, V = t.shape[0], t.shape[2]
    with K_.test_hooks() as lib:                # the TEST build of the library: the product one has no asd_debug_* switches
        try:
            lib.asd_debug_residual_groups(int(g))
            samp = K_.ResidualSampler(B, V, t.dtype)
            got = samp(t, d, n_acc, r, bo, inv_t, d_threshold=thr)
            torch.cuda.synchronize()
            return got.cpu().numpy(), samp
        finally:
            lib.asd_debug_residual_groups(0)


@pytest.mark.parametrize("dtype", [O.DT_BF16, O.DT_F32, O.DT_F16])
@pytest.mark.parametrize("B,K,V", [(32, 4, 152064), (8, 8, 152064), (5, 2, 40000), (64, 2, 32000)])
def test_residual_group_kernel_matches_the_oracle_whatever_the_workgroups_per_sequence(dtype, B, K, V):
    """asd_residual_sample_ex for few sequences: the target and the draft row of every sequence spread over 1 ... 32 workgroups
    (forced through asd_debug_residual_groups), the multi-launch form (-1) beside them: every form picks the oracle's token
    wherever the draw is >= 1e-5 of the mass from a CDF edge, with and without a nucleus-truncated draft row, bonus draws
    included; the group forms agree with each other bit for bit (canonical tile pairs, fixed fold order)."""
    import torch
    from asd_amd import kernels as K_
    rng = np.random.default_rng(B * 31 + K + V)
    if dtype == O.DT_F32 and V > 100000:
        K = min(K, 2)
    xt = (rng.standard_normal((B * K, V)) * 3).astype(np.float32)
    xd = (xt + rng.standard_normal((B * K, V))).astype(np.float32)
    st, sd = encode_logits(xt, dtype), encode_logits(xd, dtype)
    sb = encode_logits((rng.standard_normal((B, V)) * 3).astype(np.float32), dtype)
    del xt, xd
    n_acc = rng.integers(0, K + 1, B).astype(np.int32)
    r = rng.uniform(0, 1, B).astype(np.float32)
    inv_t = float(np.float32(1 / 0.7))
    thr = O.draft_sample(sd, dtype, rng.uniform(0, 1, B * K).astype(np.float32), B * K, V, inv_t, 0.9)["thr"].reshape(B, K)
    t = to_device_logits(st, dtype).view(B, K, V)
    d = to_device_logits(sd, dtype).view(B, K, V)
    bo = to_device_logits(sb, dtype).view(B, V)
    na, rr = torch.from_numpy(n_acc).cuda(), torch.from_numpy(r).cuda()
    for use_thr in (False, True):
        th = torch.from_numpy(thr).cuda() if use_thr else None
        want, margin = O.residual_sample(st, sd, dtype, n_acc, r, B, K, V, bonus=sb, inv_temperature=inv_t,
                                         d_threshold=thr if use_thr else None)
        ok = margin > 1e-5
        assert ok.mean() > 0.5
        first = None
        for g in (-1, 1, 2, 4, 8, 16, 32):
            if g > 0 and B * g > 256:
                continue
            got, samp = _residual_with_groups(K_, g, t, d, bo, na, rr, inv_t, th)
            assert np.array_equal(got[ok], want[ok]), (g, use_thr)
            nvec = V * t.element_size() // 16
            legacy = 256 + -(-B * 32 * 16 // 256) * 256 + -(-B * ((nvec + 63) // 64) * 8 // 256) * 256     # status block + scratch of the multi-launch form
            assert int(samp.buf[legacy:].count_nonzero()) == 0, "the mailboxes are handed back empty"
            assert samp.status() == 0
            n_tiles = (nvec + 63) // 64
            if g > 0 and -(-(-(-n_tiles // g)) // 16) <= 5:          # the rows fit the registers of g workgroups: a group form ran
                first = got if first is None else first
                assert np.array_equal(got, first), (g, "group forms must agree bit for bit")


def test_residual_group_kernel_edge_rows():
    """All drafted tokens accepted without a bonus row (token -1), an empty residual (p_t <= p_d everywhere: falls back to p_t),
    -inf logits over whole workgroups' runs, r = 0."""
    import torch
    from asd_amd import kernels as K_
    B, K, V = 6, 2, 65536
    rng = np.random.default_rng(9)
    xt = (rng.standard_normal((B * K, V)) * 5).astype(np.float32)      # peaked rows: the drawn tokens carry > 2e-5 of the mass
    xd = (xt + rng.standard_normal((B * K, V)) * 1.0).astype(np.float32)
    n_acc = np.array([0, 0, 2, 1, 0, 1], np.int32)          # sequence b draws from row b * K + n_acc[b]; sequence 2: all accepted
    xd[2] = xt[2]                                           # sequence 1: identical rows, the residual is empty
    xt[7, :40000] = -np.inf                                 # sequence 3: whole workgroups' runs carry no mass
    xd[7, :40000] = -np.inf
    r = rng.uniform(0, 1, B).astype(np.float32)
    r[4] = 0.0
    st, sd = encode_logits(xt, O.DT_F32), encode_logits(xd, O.DT_F32)
    t = to_device_logits(st, O.DT_F32).view(B, K, V)
    d = to_device_logits(sd, O.DT_F32).view(B, K, V)
    want, margin = O.residual_sample(st, sd, O.DT_F32, n_acc, r, B, K, V)
    for g in (0, 4, 32):
        got, _ = _residual_with_groups(K_, g, t, d, None, torch.from_numpy(n_acc).cuda(), torch.from_numpy(r).cuda(), 1.0)
        ok = (margin > 1e-5) & (want >= 0)
        assert ok[[0, 1, 3, 5]].all() and np.array_equal(got[ok], want[ok]), g
        assert got[2] == -1 and got[3] >= 40000 and 0 <= got[4] < V
