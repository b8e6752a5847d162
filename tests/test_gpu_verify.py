"""Parity of the HIP verify+accept path (through the C ABI) against the CPU oracle.

Bars (BASELINE.json): accept mask / n_acc / ballot word bit-exact on the margin-filtered set,
log-probs within atol 1e-5 + rtol 1e-6.  The accept test itself has no reference symbol
(SURVEY.md F2): parity unpinned for A5, the log-prob half is pinned to A6 via the golden
logprob_idiom.npz.
"""
import numpy as np
import pytest

from oracle import oracle as O
from tests.helpers import (LP_ATOL, assert_verify_matches, make_verify_case, run_gpu_verify, to_device_logits)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def K_():
    from asd_amd import kernels
    return kernels


@pytest.mark.parametrize("dtype", [O.DT_BF16, O.DT_F32, O.DT_F16])
def test_c1_plumbing_case(dtype):
    """BASELINE configs[0] shape: batch 1, draft_len 4, vocab 1k."""
    case = make_verify_case(1, 4, 1000, dtype, seed=1)
    assert_verify_matches(run_gpu_verify(case), case["ref"])


@pytest.mark.parametrize("dtype", [O.DT_BF16, O.DT_F32, O.DT_F16])
@pytest.mark.parametrize("V,ld", [(1001, 1001), (1000, 1003), (7, 7), (1, 5), (4099, 4101), (33, 40)])
def test_ragged_and_misaligned_rows(dtype, V, ld):
    """Odd vocab sizes and row strides: rows start off 16-byte alignment, head/tail scalars run."""
    case = make_verify_case(3, 5, V, dtype, seed=V + ld, ld_row=ld)
    assert_verify_matches(run_gpu_verify(case), case["ref"])


@pytest.mark.parametrize("B", [8, 32, 128])
def test_full_size_bf16(B):
    """BASELINE configs[1] (B=8), the headline config (B=32) and the per-node batch of configs[4]
    (B=128), draft_len 8, vocab 152064."""
    case = make_verify_case(B, 8, 152064, O.DT_BF16, seed=1234, n_threads=16)
    got = run_gpu_verify(case)
    assert_verify_matches(got, case["ref"])
    err = np.abs(got["lp_t"].astype(np.float64) - case["ref"]["lp_t64"]).max()
    print(f"B={B}: max |lp_gpu - lp_oracle| = {err:.3e}")
    assert err < LP_ATOL


def test_full_size_f32_and_f16():
    for dtype in (O.DT_F32, O.DT_F16):
        case = make_verify_case(4, 8, 152064, dtype, seed=99, n_threads=16)
        assert_verify_matches(run_gpu_verify(case), case["ref"])


@pytest.mark.parametrize("threads", [256, 512, 1024])
@pytest.mark.parametrize("unroll", [2, 4, 8])
def test_every_geometry_agrees(threads, unroll):
    """All launch geometries selectable through asd_verify_options give the oracle's answer."""
    case = make_verify_case(4, 8, 50000, O.DT_BF16, seed=5)
    for splits in (1, 2, 3, 7, 16):
        for nt in (0, 1):
            got = run_gpu_verify(case, splits=splits, threads=threads, unroll=unroll, nontemporal=nt)
            assert_verify_matches(got, case["ref"])


def test_geometry_limits_are_reported(K_):
    import torch
    case = make_verify_case(2, 64, 640, O.DT_BF16, seed=3)
    assert_verify_matches(run_gpu_verify(case), case["ref"])           # K = 64 = ASD_MAX_DRAFT_LEN
    with pytest.raises(K_.B.AsdError):                                  # K*S > 1024 staged granules
        run_gpu_verify(case, splits=32)
    lg = torch.zeros((1, 65, 16), dtype=torch.bfloat16, device="cuda")
    ws = K_.VerifyWorkspace(1, 64, 16)
    with pytest.raises(K_.B.AsdError):                                  # K > 64
        K_.verify_accept(lg, torch.zeros((1, 65), dtype=torch.int32, device="cuda"),
                         torch.zeros((1, 65), device="cuda"), torch.ones((1, 65), device="cuda"), ws)
    with pytest.raises(ValueError):                                     # no CPU path
        K_.verify_accept(lg.cpu(), torch.zeros((1, 65), dtype=torch.int32), torch.zeros((1, 65)),
                         torch.ones((1, 65)), ws)


def test_deterministic_and_workspace_reuse(K_):
    """Tickets are reset by the last arriver: one workspace, many calls, bit-identical outputs."""
    import torch
    case = make_verify_case(16, 8, 32000, O.DT_BF16, seed=11)
    lg = to_device_logits(case["logits"], case["dtype"]).view(16, 8, 32000)
    ws = K_.VerifyWorkspace(16, 8, 32000)
    tok = torch.from_numpy(case["tok"]).cuda()
    lp_d = torch.from_numpy(case["lp_d"]).cuda()
    u = torch.from_numpy(case["u"]).cuda()
    first = None
    for it in range(40):
        r = K_.verify_accept(lg, tok, lp_d, u, ws)
        cur = (r.lp_target.clone(), r.accept.clone(), r.n_acc.clone(), r.accept_bits.clone())
        if first is None:
            first = cur
        else:
            for a, b in zip(first, cur):
                assert torch.equal(a, b), it
    torch.cuda.synchronize()
    assert int(ws.buf[: 16 * 128].view(torch.int32).abs().sum()) == 0    # ticket / ballot lines back to zero
    got = dict(lp_t=first[0].cpu().numpy(), accept=first[1].cpu().numpy(), n_acc=first[2].cpu().numpy(),
               bits=first[3].cpu().numpy().view(np.uint64))
    assert_verify_matches(got, case["ref"])
    # a smaller batch reuses the same workspace
    r = K_.verify_accept(lg[:3, :5].contiguous(), tok[:3, :5].contiguous(), lp_d[:3, :5].contiguous(),
                         u[:3, :5].contiguous(), ws)
    assert np.array_equal(r.accept.cpu().numpy(), case["ref"]["accept"][:3, :5])


def test_edge_cases_follow_the_oracle():
    """Out-of-range tokens, u = 0, masked (-inf) vocabulary, +inf, NaN and all -inf rows."""
    B, K, V = 4, 6, 3000
    rng = np.random.default_rng(21)
    x = (rng.standard_normal((B * K, V)) * 4).astype(np.float32)
    x[1, 100:2900] = -np.inf            # top-p style mask
    x[2, :] = -np.inf                   # nothing survives
    x[3, 17] = np.inf
    x[4, 5] = np.nan
    x[5, :8] = -np.inf                  # a whole 16-byte vector of -inf at the row start
    tok = rng.integers(0, V, (B, K)).astype(np.int32)
    tok.reshape(-1)[0] = -1             # below the vocabulary
    tok.reshape(-1)[1] = V              # above the vocabulary
    tok.reshape(-1)[3] = 17             # the +inf logit itself
    lp_d = -np.abs(rng.standard_normal((B, K))).astype(np.float32)
    lp_d[3, 0] = -np.inf                # draft probability 0
    u = rng.uniform(0.05, 1, (B, K)).astype(np.float32)
    u[0, 0] = 0.0
    u[3, 1] = 0.0
    u[1, 0] = 0.0                       # row 6: drafted token inside the -inf mask? (set below) -> rejected at u = 0
    tok[1, 0] = 150                     # x[6, :] is unmasked; make row 6's token sit on a -inf logit instead:
    x[6, 150] = -np.inf
    for dtype in (O.DT_F32, O.DT_BF16, O.DT_F16):
        from tests.helpers import encode_logits
        store = encode_logits(x, dtype)
        ref = O.verify_accept(store, dtype, tok, lp_d, u, B, K, V)
        case = dict(B=B, K=K, V=V, dtype=dtype, ld=V, logits=store, tok=tok, lp_d=lp_d, u=u)
        with np.errstate(all="ignore"):
            got = run_gpu_verify(case)
            # rows whose margin is tiny are not part of the bit-parity set
            ok = ~(ref["margin"] < 1e-4)
            assert_verify_matches(got, ref, check_mask=False)
            assert np.array_equal(got["accept"][ok], ref["accept"][ok])
        # u = 0 with a token of ZERO target probability (id outside the vocabulary -> lp_t = -inf): rejected -- the rule
        # is u < p_t / p_d, and log u = -inf <= -inf must not accept (torch.rand draws from [0, 1))
        assert got["accept"].reshape(-1)[0] == 0 and np.isneginf(got["lp_t"].reshape(-1)[0])
        # u = 0 with a token of non-zero target probability: always accepted
        assert got["accept"][3, 1] == 1 and np.isfinite(got["lp_t"][3, 1])
        assert got["accept"][1, 0] == 0 and np.isneginf(got["lp_t"][1, 0])      # -inf target logit, u = 0
        assert np.isnan(got["lp_t"].reshape(-1)[2]) and got["accept"].reshape(-1)[2] == 0
        assert np.isnan(got["lp_t"].reshape(-1)[4]) and got["accept"].reshape(-1)[4] == 0


def test_empty_batch_is_a_noop(K_):
    import torch
    ws = K_.VerifyWorkspace(1, 1, 16)
    lg = torch.zeros((0, 4, 16), dtype=torch.bfloat16, device="cuda")
    r = K_.verify_accept(lg, torch.zeros((0, 4), dtype=torch.int32, device="cuda"), torch.zeros((0, 4), device="cuda"),
                         torch.zeros((0, 4), device="cuda"), ws)
    assert r.n_acc.numel() == 0


def test_unfiltered_set_mismatches_only_inside_margin():
    """No margin filter: any disagreement must sit within 1e-5 of the decision boundary."""
    rng = np.random.default_rng(77)
    case = make_verify_case(32, 8, 20000, O.DT_BF16, seed=77, margin=0.0)
    got = run_gpu_verify(case)
    diff = got["accept"] != case["ref"]["accept"]
    print("unfiltered mismatches:", int(diff.sum()), "of", diff.size)
    assert (case["ref"]["margin"][diff] < LP_ATOL).all()


def test_token_logprob_idiom_golden(golden, K_):
    """A6: the reference's own log(softmax(score)[tok]) values (torch f32, golden fixture)."""
    import torch
    g = golden.npz("logprob_idiom.npz")
    R, V = g["scores"].shape
    lg = torch.from_numpy(g["scores"]).cuda().view(R, 1, V)
    ws = K_.VerifyWorkspace(R, 1, V, torch.float32)
    r = K_.verify_accept(lg, torch.from_numpy(g["tok"]).cuda().view(R, 1), torch.zeros((R, 1), device="cuda"),
                         torch.full((R, 1), 0.5, device="cuda"), ws)
    np.testing.assert_allclose(r.lp_target.cpu().numpy()[:, 0], g["logprob"], rtol=1e-6, atol=1e-5)


# ---------------------------------------------------------------- size-independent properties
def test_shift_and_roll_invariance_full_width(K_):
    """log-softmax is invariant to adding a constant to a row and to rotating the vocabulary
    (with the token id rotated along); checked at the full 152064 width in f32."""
    import torch
    B, K, V = 2, 8, 152064
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn((B, K, V), generator=g, device="cuda") * 4
    tok = torch.randint(0, V, (B, K), generator=g, device="cuda", dtype=torch.int32)
    lp_d = torch.zeros((B, K), device="cuda")
    u = torch.full((B, K), 0.5, device="cuda")
    ws = K_.VerifyWorkspace(B, K, V, torch.float32)
    base = K_.verify_accept(x, tok, lp_d, u, ws).lp_target.clone()
    shifted = K_.verify_accept(x + 3.25, tok, lp_d, u, ws).lp_target.clone()
    rolled = K_.verify_accept(torch.roll(x, 1001, dims=2).contiguous(), ((tok + 1001) % V).to(torch.int32), lp_d, u,
                              ws).lp_target.clone()
    assert (base - shifted).abs().max().item() < 2e-5
    assert (base - rolled).abs().max().item() < 2e-5


def test_probabilities_of_a_row_sum_to_one(K_):
    """Verify every token id of one 4096-wide row: sum_v exp(lp_t[v]) == 1."""
    import torch
    V = 4096
    g = torch.Generator(device="cuda").manual_seed(9)
    row = (torch.randn((V,), generator=g, device="cuda") * 4).to(torch.bfloat16)
    lg = row.expand(64, 64, V).contiguous()
    toks = torch.arange(0, V, device="cuda", dtype=torch.int32).view(64, 64)
    ws = K_.VerifyWorkspace(64, 64, V)
    r = K_.verify_accept(lg, toks, torch.zeros((64, 64), device="cuda"), torch.zeros((64, 64), device="cuda"), ws)
    total = torch.exp(r.lp_target.double()).sum().item()
    assert abs(total - 1.0) < 1e-4
    assert int(r.accept.sum()) == V and bool((r.n_acc == 64).all())     # u = 0 accepts every position


def test_vocab_sharded_path_matches_single_launch(K_):
    """asd_lse_partial per shard + asd_accept_from_partials == asd_verify_accept (mask bit-exact)."""
    import torch
    case = make_verify_case(8, 8, 152064, O.DT_BF16, seed=31, n_threads=16)
    B, K, V = 8, 8, 152064
    lg = to_device_logits(case["logits"], case["dtype"]).view(B, K, V)
    tok = torch.from_numpy(case["tok"]).cuda()
    lp_d = torch.from_numpy(case["lp_d"]).cuda()
    u = torch.from_numpy(case["u"]).cuda()
    for n_shards in (2, 4, 8):
        edges = [V * i // n_shards for i in range(n_shards + 1)]
        ws = K_.VerifyWorkspace(B, K, V)
        msgs = []
        for r in range(n_shards):
            shard = lg[:, :, edges[r]:edges[r + 1]]                  # strided view: ld_row = V
            msgs.append(K_.lse_partial(shard, tok, edges[r], ws))
            ref_msg = O.lse_partial(case["logits"].reshape(B * K, V)[:, edges[r]:edges[r + 1]].copy(), O.DT_BF16,
                                    case["tok"], B, K, edges[r + 1] - edges[r], edges[r])
            m = msgs[-1].cpu().numpy().astype(np.float64)
            lse_gpu = np.log(2.0) * (m[..., 0] + np.log2(m[..., 1]))
            lse_ref = ref_msg[..., 0] + np.log(ref_msg[..., 1])
            np.testing.assert_allclose(lse_gpu, lse_ref, rtol=1e-6, atol=1e-5)
            assert np.array_equal(m[..., 2], ref_msg[..., 2])          # gathered logit: exact
        out = K_.accept_from_partials(torch.stack(msgs).contiguous(), lp_d, u)
        torch.cuda.synchronize()
        got = dict(lp_t=out.lp_target.cpu().numpy(), accept=out.accept.cpu().numpy(), n_acc=out.n_acc.cpu().numpy(),
                   bits=out.accept_bits.cpu().numpy().view(np.uint64))
        assert_verify_matches(got, case["ref"])


@pytest.mark.parametrize("B,K,V", [(8, 40, 5000), (5, 64, 3001), (300, 1, 4099), (33, 8, 7777)])
def test_row_parallel_paths_beyond_the_ballot_word(B, K, V):
    """rows >= CUs with K > 32 (ticket path at S = 1), K = 1 (per-token log-prob shape of A6), and a
    batch that is not a multiple of anything."""
    case = make_verify_case(B, K, V, O.DT_BF16, seed=B * K)
    assert_verify_matches(run_gpu_verify(case), case["ref"])


def test_more_rows_than_a_16_bit_grid_dimension():
    """The launch grid is (rows, splits): rows = B*K = 72000 exceeds the 65535 limit of the y / z dimensions, so it must sit
    in x; with explicit splits the same batch runs as 72000 x 2 workgroups through the ticket path."""
    case = make_verify_case(9000, 8, 40, O.DT_BF16, seed=77)
    assert_verify_matches(run_gpu_verify(case), case["ref"])
    assert_verify_matches(run_gpu_verify(case, splits=2, threads=256, unroll=2, nontemporal=1), case["ref"])


def test_lse_partial_row_parallel_form(K_):
    """asd_lse_partial where rows >= CUs (one workgroup per row writes its triple directly)."""
    import torch
    B, K, V = 40, 8, 24000
    case = make_verify_case(B, K, V, O.DT_BF16, seed=77)
    lg = to_device_logits(case["logits"], case["dtype"]).view(B, K, V)
    tok = torch.from_numpy(case["tok"]).cuda()
    ws = K_.VerifyWorkspace(B, K, V)
    halves = [(0, 11000), (11000, V)]
    msgs = [K_.lse_partial(lg[:, :, a:b], tok, a, ws) for a, b in halves]
    out = K_.accept_from_partials(torch.stack(msgs).contiguous(), torch.from_numpy(case["lp_d"]).cuda(),
                                  torch.from_numpy(case["u"]).cuda())
    torch.cuda.synchronize()
    got = dict(lp_t=out.lp_target.cpu().numpy(), accept=out.accept.cpu().numpy(), n_acc=out.n_acc.cpu().numpy(),
               bits=out.accept_bits.cpu().numpy().view(np.uint64))
    assert_verify_matches(got, case["ref"])


def test_concurrent_calls_on_two_streams_need_two_workspaces(K_):
    """Two verify calls in flight at once (different streams), each with its own workspace."""
    import torch
    c1 = make_verify_case(32, 8, 30000, O.DT_BF16, seed=101)
    c2 = make_verify_case(32, 8, 30000, O.DT_BF16, seed=202)
    args = []
    for c in (c1, c2):
        args.append((to_device_logits(c["logits"], c["dtype"]).view(32, 8, 30000), torch.from_numpy(c["tok"]).cuda(),
                     torch.from_numpy(c["lp_d"]).cuda(), torch.from_numpy(c["u"]).cuda(),
                     K_.VerifyWorkspace(32, 8, 30000)))
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    outs = [[], []]
    for rep in range(20):
        for i in range(2):
            with torch.cuda.stream(streams[i]):
                outs[i].append(K_.verify_accept(*args[i][:4], args[i][4]))
    torch.cuda.synchronize()
    for i, c in enumerate((c1, c2)):
        for r in outs[i]:
            assert np.array_equal(r.accept.cpu().numpy(), c["ref"]["accept"])
            assert np.array_equal(r.n_acc.cpu().numpy(), c["ref"]["n_acc"])


def test_hipgraph_capture_and_replay(K_):
    """The launcher does nothing a capture forbids and the self-resetting tickets survive replay."""
    import torch
    case = make_verify_case(32, 8, 20000, O.DT_BF16, seed=314)
    lg = to_device_logits(case["logits"], case["dtype"]).view(32, 8, 20000)
    tok, lp_d, u = (torch.from_numpy(case[k]).cuda() for k in ("tok", "lp_d", "u"))
    ws = K_.VerifyWorkspace(32, 8, 20000)
    out = K_.verify_accept(lg, tok, lp_d, u, ws)            # eager warm-up (also caches the CU count)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(3):
            K_.verify_accept(lg, tok, lp_d, u, ws, out)
    for _ in range(5):
        out.accept.zero_()
        out.n_acc.fill_(-1)
        g.replay()
        torch.cuda.synchronize()
        assert np.array_equal(out.accept.cpu().numpy(), case["ref"]["accept"])
        assert np.array_equal(out.n_acc.cpu().numpy(), case["ref"]["n_acc"])


@pytest.mark.parametrize("T", [0.7, 1.0, 1.6])
@pytest.mark.parametrize("dtype", [O.DT_BF16, O.DT_F32])
def test_temperature_is_applied_inside_the_kernel(K_, T, dtype):
    """asd_verify_options.inv_temperature: the test runs on softmax(logits / T) with no scaling pass.
    (The reference samples at temperature 0.7: generate_training_data.py:110-119.)"""
    import torch
    B, K, V = 40, 8, 30011
    case = make_verify_case(B, K, V, dtype, seed=int(T * 10), scale=6.0)
    inv_t = np.float32(1.0 / T)
    ref = O.verify_accept(case["logits"], dtype, case["tok"], case["lp_d"], case["u"], B, K, V, n_threads=8,
                          inv_temperature=float(inv_t))
    lg = to_device_logits(case["logits"], dtype).view(B, K, V)
    ws = K_.VerifyWorkspace(B, K, V, lg.dtype)
    tok, lp_d, u = (torch.from_numpy(case[k]).cuda() for k in ("tok", "lp_d", "u"))
    r = K_.verify_accept(lg, tok, lp_d, u, ws, inv_temperature=float(inv_t))
    torch.cuda.synchronize()
    got = dict(lp_t=r.lp_target.cpu().numpy(), accept=r.accept.cpu().numpy(), n_acc=r.n_acc.cpu().numpy(),
               bits=r.accept_bits.cpu().numpy().view(np.uint64))
    ok = ~(ref["margin"] < 1e-4)
    assert_verify_matches(got, ref, check_mask=False)
    assert np.array_equal(got["accept"][ok], ref["accept"][ok])
    # vocab-sharded path at the same temperature
    halves = [(0, 15000), (15000, V)]
    msgs = [K_.lse_partial(lg[:, :, a:b], tok, a, ws, inv_temperature=float(inv_t)) for a, b in halves]
    out = K_.accept_from_partials(torch.stack(msgs).contiguous(), lp_d, u, inv_temperature=float(inv_t))
    np.testing.assert_allclose(out.lp_target.cpu().numpy(), ref["lp_t64"], rtol=1e-6, atol=1e-5)
    assert np.array_equal(out.accept.cpu().numpy()[ok], ref["accept"][ok])
    with pytest.raises(K_.B.AsdError):
        K_.verify_accept(lg, tok, lp_d, u, ws, inv_temperature=0.0)


def test_c_abi_status_codes(K_):
    """Raw calls through ctypes: every documented failure is a negative asd_status, nothing throws."""
    import ctypes as C
    import torch
    from asd_amd import _binding as Bd
    lib = Bd.load_library()
    B, K, V = 4, 8, 4096
    lg = torch.zeros((B, K, V), dtype=torch.bfloat16, device="cuda")
    tok = torch.zeros((B, K), dtype=torch.int32, device="cuda")
    f = torch.zeros((B, K), dtype=torch.float32, device="cuda")
    acc = torch.zeros((B, K), dtype=torch.uint8, device="cuda")
    n = torch.zeros((B,), dtype=torch.int32, device="cuda")
    bits = torch.zeros((B,), dtype=torch.int64, device="cuda")
    ws = K_.VerifyWorkspace(B, K, V)
    st = torch.cuda.current_stream().cuda_stream

    def call(logits=lg.data_ptr(), dtype=1, ld=V, tokp=tok.data_ptr(), b=B, k=K, v=V, wsp=ws.buf.data_ptr(), wsb=ws.bytes,
             lp=f.data_ptr()):
        return lib.asd_verify_accept(logits, dtype, ld, tokp, f.data_ptr(), f.data_ptr(), b, k, v, lp, acc.data_ptr(),
                                     n.data_ptr(), bits.data_ptr(), wsp, wsb, st)

    assert call() == 0
    assert call(logits=None) == -1                     # ASD_ERR_INVALID_ARG
    assert call(tokp=None) == -1                       # ASD_ERR_INVALID_ARG
    assert call(lp=None) == -1
    assert call(b=-1) == -1
    assert call(ld=V - 1) == -1                        # rows would overlap
    assert call(dtype=7) == -2                         # ASD_ERR_UNSUPPORTED
    assert call(k=65, b=1) == -2
    assert call(wsb=64) == -3                          # ASD_ERR_WORKSPACE (too small)
    assert call(wsp=ws.buf.data_ptr() + 8) == -3       # misaligned
    assert call(logits=lg.data_ptr() + 1) == -5        # ASD_ERR_ALIGNMENT (odd address for a 2-byte type)
    assert call(b=0) == 0 and call(k=0) == 0           # empty problems are no-ops
    d = torch.zeros((4, 3), dtype=torch.float64, device="cuda")
    ks = torch.zeros((4,), dtype=torch.int32, device="cuda")
    assert lib.asd_optimal_stopping(d.data_ptr(), d.data_ptr(), 1.0, 4, 0, 0, 1.0, 1.0, ks.data_ptr(), None, st) == -1
    assert lib.asd_optimal_stopping(d.data_ptr(), d.data_ptr(), 1.0, 4, 17, 0, 1.0, 1.0, ks.data_ptr(), None, st) == -2
    assert lib.asd_optimal_stopping(None, d.data_ptr(), 1.0, 4, 3, 0, 1.0, 1.0, ks.data_ptr(), None, st) == -1
    assert lib.asd_mlp_predict(f.data_ptr(), 8, f.data_ptr(), 4, 2000, 32, f.data_ptr(), st) == -2
    assert lib.asd_logprob_stats(f.data_ptr(), 4, None, 4, 8, d.data_ptr(), st) == -1     # ld < K
    assert lib.asd_workspace_init(None, 256, st) == -1
    torch.cuda.synchronize()
    assert call() == 0                                 # the library is still healthy afterwards


def test_fuzz_shapes_strides_dtypes_geometries(K_):
    """60 seeded random problems (ASD_FUZZ_CASES / ASD_FUZZ_SEED override the count and the seed for long soaks): batch, draft
    length, vocabulary (incl. tiny and odd), row padding, dtype, temperature and launch geometry all drawn at random; every
    one must match the oracle."""
    import os

    import torch
    rng = np.random.default_rng(int(os.environ.get("ASD_FUZZ_SEED", "20251004")))
    fuzz_epi = {}
    for it in range(int(os.environ.get("ASD_FUZZ_CASES", "60"))):
        B = int(rng.integers(1, 70))
        K = int(rng.choice([1, 2, 3, 4, 5, 8, 13, 16, 32, 33, 64]))
        if B * K > 1200:
            B = max(1, 1200 // K)
        V = int(rng.choice([1, 2, 7, 8, 9, 63, 64, 65, 511, 512, 513, 1000, 4096, 5003, 20000, 32768, 70001]))
        dtype = int(rng.choice([O.DT_BF16, O.DT_F32, O.DT_F16]))
        ld = V + int(rng.choice([0, 0, 1, 3, 8, 17]))
        inv_t = float(np.float32(rng.choice([1.0, 1.0, 1.0 / 0.7, 0.5, 2.0])))
        case = make_verify_case(B, K, V, dtype, seed=1000 + it, ld_row=ld, scale=float(rng.choice([1.0, 4.0, 8.0])))
        ref = O.verify_accept(case["logits"], dtype, case["tok"], case["lp_d"], case["u"], B, K, V, ld_row=ld,
                              n_threads=8, inv_temperature=inv_t)
        geom = {}
        if rng.uniform() < 0.5:
            smax = max(1, min(64, 1024 // K))
            geom = dict(splits=int(rng.integers(1, smax + 1)), threads=int(rng.choice([256, 512, 1024])),
                        unroll=int(rng.choice([2, 3, 4, 8])), nontemporal=int(rng.integers(0, 2)))
        lg = to_device_logits(case["logits"], dtype)
        lg3 = lg.as_strided((B, K, V), (K * ld, ld, 1))
        ws = K_.VerifyWorkspace(B, K, V, lg.dtype)
        r = K_.verify_accept(lg3, torch.from_numpy(case["tok"]).cuda(), torch.from_numpy(case["lp_d"]).cuda(),
                             torch.from_numpy(case["u"]).cuda(), ws, inv_temperature=inv_t, **geom)
        torch.cuda.synchronize()
        got = dict(lp_t=r.lp_target.cpu().numpy(), accept=r.accept.cpu().numpy(), n_acc=r.n_acc.cpu().numpy(),
                   bits=r.accept_bits.cpu().numpy().view(np.uint64))
        ok = ~(ref["margin"] < 1e-4)
        try:
            assert_verify_matches(got, ref, check_mask=False)
            assert np.array_equal(got["accept"][ok], ref["accept"][ok])
            if ok.all():
                assert np.array_equal(got["n_acc"], ref["n_acc"]) and np.array_equal(got["bits"], ref["bits"])
            if not geom:     # the one-launch step on the heuristic's geometry: same verify outputs bit for bit, workspace handed back empty
                feat, packed, Cc = fuzz_epi.setdefault("args", _fused_args(K_, 70))
                ph = torch.ones((B, 3), dtype=torch.float64, device="cuda")
                v2, s2 = K_.verify_accept_fused(lg3, torch.from_numpy(case["tok"]).cuda(), torch.from_numpy(case["lp_d"]).cuda(),
                                                torch.from_numpy(case["u"]).cuda(), ws, feat[:B].contiguous(), packed, 64, 32,
                                                stage_idx=1, L=3, p_hist=ph, Cc=Cc, lam=0.8, inv_temperature=inv_t)
                torch.cuda.synchronize()
                assert torch.equal(v2.lp_target, r.lp_target) or bool(torch.isnan(r.lp_target).any())
                assert torch.equal(v2.accept, r.accept) and torch.equal(v2.n_acc, r.n_acc) and torch.equal(v2.accept_bits, r.accept_bits)
                assert bool(torch.isfinite(s2.score).all()) or bool(torch.isnan(r.lp_target).any())
                assert int(ws.buf.count_nonzero()) == 0
        except AssertionError as e:
            raise AssertionError(f"fuzz case {it}: B={B} K={K} V={V} ld={ld} dtype={dtype} inv_t={inv_t} geom={geom}") from e


def test_reference_idiom_goldens_at_full_vocabulary(golden):
    """A6 pinned where the path runs (V = 152064): asd_verify_accept_ex against what the reference's own loop
    (generate_training_data.py:128-136, executed by oracle/gen_golden.py in the dev container) returns for scores
    that went through bf16 / fp16 storage, were divided by T = 0.7 and masked to the top-p = 0.9 nucleus by HF's
    warpers (:110-119).  Inputs are regenerated from the fixture's seeds.  Two forms per 16-bit row: the raw stored
    logits with the temperature folded into the pass, and the processed f32 row as the reference loop saw it."""
    import torch
    from asd_amd import kernels as K
    from tests.helpers import LP_RTOL, REF_F32_SUM_ERR, encode_logits, full_size_cases, to_device_logits
    g = golden.npz("logprob_idiom_full.npz")
    T = np.float32(g["temperature"])
    inv_t = float(np.float32(1.0) / T)
    groups = {}
    for var, tok, want, x, keep in full_size_cases(g):
        V = x.size
        if var == "f32":
            groups.setdefault(("f32", O.DT_F32, 1.0), []).append((x, tok, want))
            continue
        dt = O.DT_BF16 if var.startswith("bf16") else O.DT_F16
        xs = x.copy()
        if keep is not None:
            mask = np.ones(V, bool)
            mask[keep] = False
            xs[mask] = -np.inf
        groups.setdefault((var + " stored, T in-kernel", dt, inv_t), []).append((xs, tok, want))
        groups.setdefault((var + " processed f32 row", O.DT_F32, 1.0), []).append(((xs / T).astype(np.float32), tok, want))
    checked = 0
    for (name, dt, it), rows in groups.items():
        n = len(rows)
        store = np.stack([encode_logits(r[0], dt) for r in rows])
        lg = to_device_logits(store, dt).view(n, 1, V)
        tok = torch.tensor([[r[1]] for r in rows], dtype=torch.int32, device="cuda")
        ws = K.VerifyWorkspace(n, 1, V, lg.dtype)
        res = K.verify_accept(lg, tok, torch.zeros((n, 1), device="cuda"), torch.full((n, 1), 0.5, device="cuda"), ws,
                              inv_temperature=it)
        torch.cuda.synchronize()
        got = res.lp_target.cpu().numpy()[:, 0].astype(np.float64)
        want = np.array([r[2] for r in rows])
        fin = np.isfinite(want)
        # (1) against the exact value: the f64 oracle on the same stored row, BASELINE's 1e-5
        exact = O.verify_accept(store, dt, tok.cpu().numpy(), np.zeros(n, np.float32), np.full(n, 0.5, np.float32), n, 1, V,
                                inv_temperature=np.float32(it))["lp_t64"][:, 0]
        np.testing.assert_allclose(got[fin], exact[fin], rtol=LP_RTOL, atol=LP_ATOL, err_msg=name)
        # (2) against the reference's own f32 idiom: its softmax sum carries up to ~1.7e-5 of f32 accumulation error
        # on full-vocabulary rows (helpers.REF_F32_SUM_ERR); on nucleus rows it is exact and so must the kernel be
        tol = 2e-6 if "_topp" in name else REF_F32_SUM_ERR
        np.testing.assert_allclose(got[fin], want[fin], rtol=LP_RTOL, atol=tol, err_msg=name)
        assert np.isneginf(got[~fin]).all(), name
        checked += n
    assert checked == 6 + 4 * 12


@pytest.mark.parametrize("dtype", [O.DT_BF16, O.DT_F32, O.DT_F16])
@pytest.mark.parametrize("B,K,V,ld", [(32, 8, 152064, None), (3, 5, 1003, 1010), (8, 8, 32000, None), (2, 32, 4096, None)])
def test_verify_stats_emit_max_logprob_and_entropy(K_, dtype, B, K, V, ld):
    """asd_verify_accept_stats (N1, the A14 half): the verify outputs are those of asd_verify_accept_ex bit for bit (same geometry), and
    per position max log-prob / softmax entropy match the f64 oracle (doc-only quantities: tolerance 2e-5 / 1e-4)."""
    import torch
    case = make_verify_case(B, K, V, dtype, seed=V + B, ld_row=ld)
    x = O.logits_as_f32(case["logits"], dtype).copy()
    if V > 2000:
        x[1, 100:V - 100] = -np.inf                      # a top-p style masked row: entropy over the survivors only
        from tests.helpers import encode_logits
        case["logits"] = encode_logits(x, dtype)
        case["ref"] = O.verify_accept(case["logits"], dtype, case["tok"], case["lp_d"], case["u"], B, K, V, ld_row=case["ld"])
    inv_t = float(np.float32(1 / 0.7))
    lg = to_device_logits(case["logits"], dtype)
    lg3 = lg.as_strided((B, K, V), (K * case["ld"], case["ld"], 1))
    ws = K_.VerifyWorkspace(B, K, V, lg.dtype)
    tok, lp_d, u = (torch.from_numpy(case[k]).cuda() for k in ("tok", "lp_d", "u"))
    # the entropy form runs one 512-lane workgroup per row with 3-KiB tiles: bit-identical to the plain kernel AT THAT geometry
    plain = K_.verify_accept(lg3, tok, lp_d, u, ws, inv_temperature=inv_t, splits=1, threads=512, unroll=3, nontemporal=1)
    res, max_lp, ent = K_.verify_accept_stats(lg3, tok, lp_d, u, ws, inv_temperature=inv_t)
    only_max = K_.verify_accept_stats(lg3, tok, lp_d, u, ws, inv_temperature=inv_t, want_entropy=False)
    torch.cuda.synchronize()
    for a, b in ((res.lp_target, plain.lp_target), (res.accept, plain.accept), (res.n_acc, plain.n_acc), (res.accept_bits, plain.accept_bits)):
        assert torch.equal(a, b)
    want_max, want_ent = O.row_softmax_stats(case["logits"], dtype, B * K, V, inv_t, ld_row=case["ld"])
    got_max, got_ent = max_lp.cpu().numpy().reshape(-1).astype(np.float64), ent.cpu().numpy().reshape(-1).astype(np.float64)
    np.testing.assert_allclose(got_max, want_max, atol=2e-5, rtol=0)
    np.testing.assert_allclose(got_ent, want_ent, atol=1e-4, rtol=1e-5)
    np.testing.assert_allclose(only_max[1].cpu().numpy().reshape(-1), want_max, atol=2e-5, rtol=0)
    assert only_max[2] is None
    assert (got_ent >= -1e-5).all() and (got_ent <= np.log(V) + 1e-4).all()


def test_verify_stats_limits(K_):
    import torch
    B, K, V = 2, 40, 512                                   # K > 32: the entropy form is one-workgroup-per-row, ballot-by-atomic only
    lg = torch.zeros((B, K, V), dtype=torch.bfloat16, device="cuda")
    ws = K_.VerifyWorkspace(B, K, V)
    z = torch.zeros((B, K), device="cuda")
    with pytest.raises(K_.B.AsdError):
        K_.verify_accept_stats(lg, torch.zeros((B, K), dtype=torch.int32, device="cuda"), z, z + 0.5, ws)
    r, mx, ent = K_.verify_accept_stats(lg, torch.zeros((B, K), dtype=torch.int32, device="cuda"), z, z + 0.5, ws, want_entropy=False)
    torch.cuda.synchronize()
    np.testing.assert_allclose(mx.cpu().numpy(), -np.log(V), atol=1e-5)      # a flat row: every log-prob is -ln V


def _fused_args(K_, B, seed=5):
    import torch
    from tests.conftest import load_npz
    g = load_npz("predictor.npz")
    packed = K_.pack_mlp_weights(g["w1"], g["b1"], g["w2"], g["b2"])
    rng = np.random.default_rng(seed)
    feat = torch.from_numpy((rng.standard_normal((B, 64)) * 0.3).astype(np.float32)).cuda()
    Cc = torch.tensor([1.0, 4.5, 10.0], dtype=torch.float64, device="cuda")
    return feat, packed, Cc


def test_workspace_is_all_zero_after_every_call_form_and_shared_across_shapes(K_):
    """The workspace invariant: tickets, ballot words, lp slots and granule slots are zero between
    calls, so calls of DIFFERENT shapes may share one workspace in stream order.  The layout depends on B (the granule
    regions start behind B ticket blocks): a fused call at B = 32 followed by a split call at B = 8 puts granule regions
    where the lp hand-off slots were -- a stale non-zero word there would be taken for a published granule."""
    import torch
    V = 30000
    big = make_verify_case(32, 8, V, O.DT_BF16, seed=41)
    small = make_verify_case(8, 8, V, O.DT_BF16, seed=42)
    dev = {}
    for name, c in (("big", big), ("small", small)):
        dev[name] = (to_device_logits(c["logits"], c["dtype"]).view(c["B"], c["K"], V),) + tuple(
            torch.from_numpy(c[k]).cuda() for k in ("tok", "lp_d", "u"))
    ws = K_.VerifyWorkspace(32, 8, V)
    fresh = K_.verify_accept(*dev["small"], K_.VerifyWorkspace(8, 8, V))
    torch.cuda.synchronize()

    def clean():
        torch.cuda.synchronize()
        assert int(ws.buf.count_nonzero()) == 0

    def check(r, c):
        assert np.array_equal(r.accept.cpu().numpy(), c["ref"]["accept"]) and np.array_equal(r.n_acc.cpu().numpy(), c["ref"]["n_acc"])

    feat32, packed, Cc = _fused_args(K_, 32)
    feat8 = feat32[:8].contiguous()
    for rep in range(3):
        ph = torch.ones((32, 3), dtype=torch.float64, device="cuda")
        v, s = K_.verify_accept_fused(*dev["big"], ws, feat32, packed, 64, 32, stage_idx=0, L=3, p_hist=ph, Cc=Cc, lam=0.8)
        clean(); check(v, big)                                                           # fused, one workgroup per row
        r = K_.verify_accept(*dev["small"], ws)                                          # split rows, B = 8, same workspace
        clean(); check(r, small)
        assert torch.equal(r.lp_target, fresh.lp_target) and torch.equal(r.accept_bits, fresh.accept_bits)
        ph8 = torch.ones((8, 3), dtype=torch.float64, device="cuda")
        v8, s8 = K_.verify_accept_fused(*dev["small"], ws, feat8, packed, 64, 32, stage_idx=0, L=3, p_hist=ph8, Cc=Cc, lam=0.8)
        clean(); check(v8, small)                                                        # fused, split rows
        assert torch.equal(v8.lp_target, fresh.lp_target)
        r = K_.verify_accept(*dev["big"], ws)                                            # plain, one workgroup per row
        clean(); check(r, big)
        r = K_.verify_accept(*dev["big"], ws, splits=3)                                  # forced split at rows >= CUs
        clean(); check(r, big)
        K_.verify_accept_stats(*dev["big"], ws)
        clean()
        K_.lse_partial(dev["small"][0], dev["small"][1], 0, ws)
        clean()
        K_.lse_partial(dev["big"][0], dev["big"][1], 0, ws)
        clean()


@pytest.mark.parametrize("form", ["split", "fused_rows", "fused_split", "fused_rows_256x128"])
def test_a_withheld_hand_off_poisons_the_result_instead_of_folding_a_zero(K_, form):
    """asd_debug_verify_withhold makes one workgroup skip its hand-off store: the finisher's bounded wait must end in
    NaN / reject (split rows) or score = NaN, k* = L - 1, stop = 0 (in-kernel epilogue) for THAT row / sequence only,
    and leave the workspace clean for the next call."""
    import torch
    V = 40000                                               # wide enough for ceil(CUs / rows) = 4 slices of >= one tile per wave
    B = 32 if form.startswith("fused_rows") else 8
    c = make_verify_case(B, 8, V, O.DT_BF16, seed=B + 7)
    lg = to_device_logits(c["logits"], c["dtype"]).view(B, 8, V)
    tok, lp_d, u = (torch.from_numpy(c[k]).cuda() for k in ("tok", "lp_d", "u"))
    ws = K_.VerifyWorkspace(B, 8, V)
    feat, packed, Cc = _fused_args(K_, B)
    dims = (64, 32)
    if form == "fused_rows_256x128":                        # the eight-wave epilogue (k_verify<..., EPI = 2>): all waves of the finisher's
        rng = np.random.default_rng(3)                      # workgroup meet at a barrier BEFORE the loss is detected
        packed = K_.pack_mlp_weights((rng.standard_normal((128, 256)) / 16).astype(np.float32), (rng.standard_normal(128) * 0.1).astype(np.float32),
                                     (rng.standard_normal((1, 128)) / 8).astype(np.float32), np.array([0.05], np.float32))
        feat = torch.from_numpy((rng.standard_normal((B, 256)) * 0.3).astype(np.float32)).cuda()
        dims = (256, 128)
    cus = K_.device_cu_count()
    S = 1 if B * 8 >= cus else -(-cus // (B * 8))          # the launcher's split count (rows < CUs: ceil(CUs / rows))
    bad_b, bad_k = 3, 5
    row = bad_b * 8 + bad_k
    with K_.test_hooks() as lib:                            # the fault-injection switch exists in the TEST build of the library only
        try:
            lib.asd_debug_verify_withhold(row * S + (S - 1))
            ph = torch.ones((B, 3), dtype=torch.float64, device="cuda")
            if form == "split":
                v, s = K_.verify_accept(lg, tok, lp_d, u, ws), None
            else:
                v, s = K_.verify_accept_fused(lg, tok, lp_d, u, ws, feat, packed, *dims, stage_idx=0, L=3, p_hist=ph, Cc=Cc, lam=0.8)
            torch.cuda.synchronize()
        finally:
            lib.asd_debug_verify_withhold(-1)
    lp = v.lp_target.cpu().numpy()
    acc = v.accept.cpu().numpy()
    others = np.ones((B, 8), bool)
    if form.startswith("fused_rows"):                       # rows finish themselves: only the epilogue lost its input
        np.testing.assert_allclose(lp, c["ref"]["lp_t64"], atol=1e-5, rtol=1e-6)
        assert np.array_equal(acc, c["ref"]["accept"])
    else:
        others[bad_b, bad_k] = False
        assert np.isnan(lp[bad_b, bad_k]) and acc[bad_b, bad_k] == 0
        np.testing.assert_allclose(lp[others], c["ref"]["lp_t64"][others], atol=1e-5, rtol=1e-6)
        assert np.array_equal(acc[others], c["ref"]["accept"][others])
        assert int(v.n_acc[bad_b]) <= bad_k
    if s is not None:
        sc = s.score.cpu().numpy()
        assert np.isnan(sc[bad_b]) and int(s.k_star[bad_b]) == 2 and int(s.stop[bad_b]) == 0
        assert np.isfinite(np.delete(sc, bad_b)).all()
    # the loss is REPORTED: the workspace's sticky status word (its first 32 bits) says so, through the C ABI and as a tensor view;
    # nothing else of the workspace is left dirty by this hook (the withheld word was never written)
    assert ws.status() == K_.B.WS_LOST_HANDOFF and int(ws.status_word) == K_.B.WS_LOST_HANDOFF
    assert int(ws.buf[4:].count_nonzero()) == 0
    with pytest.raises(K_.LostHandoffError):
        ws.check()                                          # raises once and re-initialises the workspace ...
    ws.check()                                              # ... so that it is clean again
    assert int(ws.buf.count_nonzero()) == 0
    again = K_.verify_accept(lg, tok, lp_d, u, ws)         # the hook is off: the same workspace serves a clean call
    torch.cuda.synchronize()
    assert np.array_equal(again.accept.cpu().numpy(), c["ref"]["accept"])
    assert ws.status() == 0


@pytest.mark.parametrize("B,K,V", [(33, 8, 152064), (40, 8, 30000), (48, 8, 30000), (65, 8, 30000), (100, 4, 20000), (130, 8, 9000), (36, 8, 1000)])
def test_rows_that_do_not_fill_the_cus_evenly_are_cut_into_slices(K_, B, K, V):
    """rows >= CUs but not a multiple of them: the heuristic cuts the rows into 4 (2) slices so that every CU gets nearly the same
    bytes (choose_geometry).  Same results as the oracle, the one-launch step equals two launches bit for bit (both take the same
    geometry), the workspace comes back clean; the (m2, s, t) instantiation and the 256 x 128 epilogue keep whole rows."""
    import torch
    case = make_verify_case(B, K, V, O.DT_BF16, seed=B * 3 + K)
    lg = to_device_logits(case["logits"], case["dtype"]).view(B, K, V)
    tok, lp_d, u = (torch.from_numpy(case[k]).cuda() for k in ("tok", "lp_d", "u"))
    ws = K_.VerifyWorkspace(B, K, V)
    v = K_.verify_accept(lg, tok, lp_d, u, ws)
    torch.cuda.synchronize()
    assert np.array_equal(v.accept.cpu().numpy(), case["ref"]["accept"]) and np.array_equal(v.n_acc.cpu().numpy(), case["ref"]["n_acc"])
    np.testing.assert_allclose(v.lp_target.cpu().numpy(), case["ref"]["lp_t64"], atol=1e-5, rtol=1e-6)
    forced = K_.verify_accept(lg, tok, lp_d, u, ws, splits=1)                       # whole rows: the same decisions
    assert np.array_equal(forced.accept.cpu().numpy(), case["ref"]["accept"])
    feat, packed, Cc = _fused_args(K_, B)
    ph1 = torch.ones((B, 3), dtype=torch.float64, device="cuda")
    ph2 = ph1.clone()
    s1 = K_.predictor_stop(feat, packed, 64, 32, stage_idx=0, L=3, lp=v.lp_target, stats_col=5, p_hist=ph1, Cc=Cc, lam=0.8)
    v2, s2 = K_.verify_accept_fused(lg, tok, lp_d, u, ws, feat, packed, 64, 32, stage_idx=0, L=3, stats_col=5, p_hist=ph2, Cc=Cc, lam=0.8)
    torch.cuda.synchronize()
    for a, b in ((v.lp_target, v2.lp_target), (v.accept_bits, v2.accept_bits), (v.n_acc, v2.n_acc), (s1.score, s2.score), (s1.k_star, s2.k_star),
                 (s1.stop, s2.stop), (ph1, ph2)):
        assert torch.equal(a, b)
    st = K_.verify_accept_stats(lg, tok, lp_d, u, ws) if K <= 32 else None
    torch.cuda.synchronize()
    if st is not None:
        assert np.array_equal(st[0].accept.cpu().numpy(), case["ref"]["accept"])
    assert int(ws.buf.count_nonzero()) == 0 and ws.status() == 0
