"""N2: asd_lm_head_verify -- the lm_head GEMM fused with the verify pass -- against
the f64 oracle (oracle.lm_head_verify: f64 product of the bf16 operands, then the A5 rule).

Tolerance: the kernel accumulates bf16 x bf16 products in f32 MFMA accumulators over D terms, so
its logits carry ~sqrt(D) * 2^-24 * |x| of rounding that the streaming kernel (which reads given
logits) does not have.  LMH_ATOL = 2e-4 on the log-prob covers D = 8192 at |x| <~ 30; the accept
mask / n_acc / ballot word must be identical on inputs whose decision margin is >= 10 * LMH_ATOL.
Parity unpinned in the sense of DESIGN.md: A5 has no reference symbol."""
import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu

LMH_ATOL = 2e-4
LMH_MARGIN = 2e-3


def make_case(B, K, D, V, seed=0, scale=3.0, ld_h=None, ld_w=None, inv_t=1.0):
    rng = np.random.default_rng(seed)
    M = B * K
    hb = O.f32_to_bf16_bits(rng.standard_normal((M, D), dtype=np.float32))
    wb = O.f32_to_bf16_bits(rng.standard_normal((V, D), dtype=np.float32) * np.float32(scale / np.sqrt(D)))
    x = O.bf16_bits_to_f32(hb).astype(np.float64) @ O.bf16_bits_to_f32(wb).astype(np.float64).T
    amax = x.argmax(axis=1)
    tok = np.where(rng.uniform(size=M) < 0.7, amax, rng.integers(0, V, M)).astype(np.int32).reshape(B, K)
    base = O.lm_head_verify(hb, wb, tok, np.zeros((B, K), np.float32), np.full((B, K), 0.5, np.float32), B, K, inv_t)
    lp_t = base["lp_t64"]
    lp_d = np.minimum(lp_t + rng.normal(0, 0.5, (B, K)), 0.0).astype(np.float32)
    u = rng.uniform(0, 1, (B, K)).astype(np.float32)
    for _ in range(100):
        with np.errstate(divide="ignore"):
            m = np.abs(np.log(u.astype(np.float64)) - (lp_t - lp_d.astype(np.float64)))
        bad = ~(m >= LMH_MARGIN)
        if not bad.any():
            break
        u[bad] = rng.uniform(0, 1, int(bad.sum())).astype(np.float32)
    ref = O.lm_head_verify(hb, wb, tok, lp_d, u, B, K, inv_t)
    return dict(B=B, K=K, D=D, V=V, hb=hb, wb=wb, tok=tok, lp_d=lp_d, u=u, ref=ref, ld_h=ld_h or D, ld_w=ld_w or D,
                inv_t=inv_t)


def bf16_dev(bits, ld=None):
    import torch

    t = torch.from_numpy(bits.view(np.int16)).cuda().view(torch.bfloat16)
    if ld is None or ld == bits.shape[1]:
        return t
    pad = torch.full((bits.shape[0], ld), 1.0e4, dtype=torch.bfloat16, device="cuda")   # poisoned padding
    pad[:, :bits.shape[1]] = t
    return pad[:, :bits.shape[1]]


def run_gpu(case):
    import torch

    from asd_amd import kernels as Kn

    w = bf16_dev(case["wb"], case["ld_w"])
    h = bf16_dev(case["hb"], case["ld_h"])
    ver = Kn.LmHeadVerifier(w, case["B"], case["K"])
    r = ver(h, torch.from_numpy(case["tok"]).cuda(), torch.from_numpy(case["lp_d"]).cuda(),
            torch.from_numpy(case["u"]).cuda(), inv_temperature=case["inv_t"])
    torch.cuda.synchronize()
    return dict(lp_t=r.lp_target.cpu().numpy(), accept=r.accept.cpu().numpy(), n_acc=r.n_acc.cpu().numpy(),
                bits=r.accept_bits.cpu().numpy().view(np.uint64))


def check(got, ref):
    a, b = got["lp_t"].astype(np.float64), ref["lp_t64"]
    fin = np.isfinite(b)
    np.testing.assert_allclose(a[fin], b[fin], rtol=0, atol=LMH_ATOL)
    assert np.array_equal(a[~fin], b[~fin])
    assert np.array_equal(got["accept"], ref["accept"])
    assert np.array_equal(got["n_acc"], ref["n_acc"])
    assert np.array_equal(got["bits"], ref["bits"])


@pytest.mark.parametrize("B,K,D,V", [
    (1, 1, 64, 5),           # one row, one column block, mostly padding
    (3, 5, 64, 300),         # ragged rows and a ragged last column block
    (5, 8, 256, 1000),
    (32, 8, 512, 4173),      # the full 256-row tile
    (40, 8, 128, 640),       # two row blocks
    (9, 33, 192, 129),       # K > 32, 297 rows, one column in the last block, odd superstage count
    (2, 64, 320, 257),       # K = ASD_MAX_DRAFT_LEN
    (32, 8, 128, 66000),     # wide (256-column) blocks for whole rounds of CUs + narrow blocks for the rest
    (32, 8, 4096, 65636),    # ... and, for a deep reduction, the rest as a 256-column block cut into 2 reduction
    (32, 8, 4608, 65700),    # slices / 3 slices (ticketed f32 slabs); the block is ragged (100 / 164 real columns)
    (32, 8, 4608, 65536 + 3 * 256 - 56),   # THREE split-K tail tiles x 3 slices: slab index (nb*k_slices+sl)*8+wv and tickets[nb], nb > 0
    (32, 8, 4096, 65536 + 5 * 256),        # five tail tiles x 2 slices, last tile full
    (40, 8, 128, 4000),      # m_blocks = 2 with 32 narrow column blocks: the XCD-swizzled (mb, nb) map (n_blocks >= 8)
    (40, 8, 128, 33000),     # m_blocks = 2 with 128 WIDE blocks (swizzled) + 2 narrow blocks in the plain order
    (72, 8, 64, 3000),       # m_blocks = 3 (576 rows), 23 narrow blocks: 16 swizzled + 7 in the plain order
    # M > 256 runs the 4-wave kernel (k_lm_head_quad: 128 x 128 per wave, register-staged operands) over every 256-column block
    (128, 8, 256, 5000),     # M = 1024: four full row blocks, 20 column blocks (16 in the XCD-swizzled order, 4 plain, the last ragged)
    (80, 8, 192, 4037),      # M = 640: a half row block (wave row 1 of the last block is padding), three superstages, odd V
    (40, 8, 128, 700),       # two superstages (the short pipeline), three column blocks
    (37, 8, 64, 300),        # ONE superstage; 296 rows: 40 real rows in the second block; ragged columns in both blocks
    (64, 8, 512, 2304),      # M = 512, nine full column blocks
    (96, 8, 1536, 256 * 100),         # m_blocks = 3, 300 tiles: more than one round of the CUs
    # 256 < M <= 288 runs ONE tall row block (k_lm_head_tall: 1 x 8 waves of 9 x 1 tiles), eight wave columns merged through LDS
    (33, 8, 128, 300),       # M = 264: eight real rows in the ninth tile; two column blocks, the last ragged (44 columns: waves 2..7 idle)
    (36, 8, 512, 4173),      # M = 288 exactly; 17 column blocks, eight superstages
    (32, 9, 192, 70000),     # K + 1 = 9 positions of 32 sequences; 274 column blocks (more than one round of the CUs), odd superstage count
    (34, 8, 64, 257),        # ONE superstage; a single column in the second block
])
def test_lm_head_verify_matches_oracle(B, K, D, V):
    case = make_case(B, K, D, V, seed=B * 1000 + K)
    check(run_gpu(case), case["ref"])


@pytest.mark.parametrize("B,K,D,V", [
    (3, 5, 64, 300),          # skinny kernel (M <= 64)
    (8, 8, 256, 1000),
    (32, 8, 512, 4173),       # the 8-wave tile kernel, wide + narrow blocks
    (32, 8, 4608, 65536 + 3 * 256 - 56),   # ... with reduction slices
    (128, 8, 256, 5000),      # the 4-wave kernel (M > 256)
    (37, 8, 64, 300),         # ... with a partial last row block
    (35, 8, 256, 5000),       # the tall row block (256 < M <= 288)
])
def test_lm_head_verify_f16_matches_oracle(B, K, D, V):
    """The same call on f16 hidden states and weights (ASD_DTYPE_F16: the reference loads its models in fp16,
    generate_training_data.py:79-85) -- v_mfma_f32_32x32x16_f16 on the same LDS images -- through every kernel variant, plain
    and packed, against the f64 oracle on the f16 values."""
    import torch

    from asd_amd import kernels as Kn

    rng = np.random.default_rng(B * 31 + V)
    M = B * K
    hb = rng.standard_normal((M, D), dtype=np.float32).astype(np.float16).view(np.uint16)
    wb = (rng.standard_normal((V, D), dtype=np.float32) * np.float32(3.0 / np.sqrt(D))).astype(np.float16).view(np.uint16)
    x = hb.view(np.float16).astype(np.float64) @ wb.view(np.float16).astype(np.float64).T
    amax = x.argmax(axis=1)
    tok = np.where(rng.uniform(size=M) < 0.7, amax, rng.integers(0, V, M)).astype(np.int32).reshape(B, K)
    base = O.lm_head_verify(hb, wb, tok, np.zeros((B, K), np.float32), np.full((B, K), 0.5, np.float32), B, K, 1.0, dtype=O.DT_F16)
    lp_t = base["lp_t64"]
    lp_d = np.minimum(lp_t + rng.normal(0, 0.5, (B, K)), 0.0).astype(np.float32)
    u = rng.uniform(0, 1, (B, K)).astype(np.float32)
    for _ in range(100):
        with np.errstate(divide="ignore"):
            m = np.abs(np.log(u.astype(np.float64)) - (lp_t - lp_d.astype(np.float64)))
        bad = ~(m >= LMH_MARGIN)
        if not bad.any():
            break
        u[bad] = rng.uniform(0, 1, int(bad.sum())).astype(np.float32)
    ref = O.lm_head_verify(hb, wb, tok, lp_d, u, B, K, 1.0, dtype=O.DT_F16)
    w = torch.from_numpy(wb.view(np.int16)).cuda().view(torch.float16)
    h = torch.from_numpy(hb.view(np.int16)).cuda().view(torch.float16)
    args = (h, torch.from_numpy(tok).cuda(), torch.from_numpy(lp_d).cuda(), torch.from_numpy(u).cuda())
    outs = []
    for packed in (False, True):
        r = Kn.LmHeadVerifier(w, B, K, packed=packed)(*args)
        torch.cuda.synchronize()
        got = dict(lp_t=r.lp_target.cpu().numpy(), accept=r.accept.cpu().numpy(), n_acc=r.n_acc.cpu().numpy(),
                   bits=r.accept_bits.cpu().numpy().view(np.uint64))
        check(got, ref)
        outs.append(got)
    assert np.array_equal(outs[0]["lp_t"], outs[1]["lp_t"])      # the packed image gives the same bits
    with pytest.raises(ValueError):                              # operand types must agree
        Kn.LmHeadVerifier(w, B, K)(h.to(torch.bfloat16), *args[1:])


@pytest.mark.parametrize("B,K,D,V", [(3, 5, 64, 300), (8, 8, 256, 1000), (32, 8, 512, 4173), (40, 8, 128, 33000),
                                     (32, 8, 128, 66000), (32, 8, 4608, 65536 + 3 * 256 - 56), (16, 8, 192, 129),
                                     (128, 8, 256, 5000), (37, 8, 64, 300), (36, 8, 192, 3000)])
def test_packed_weights_give_bit_identical_results(B, K, D, V):
    """asd_lm_head_pack_weights: the tile-major image (ld_w = 0) through every kernel variant -- skinny (M <= 64), wide and
    narrow blocks, two row blocks, split-K tail tiles -- returns exactly what the [V, D] matrix returns."""
    import torch

    from asd_amd import kernels as Kn

    case = make_case(B, K, D, V, seed=B + V)
    w, h = bf16_dev(case["wb"]), bf16_dev(case["hb"])
    args = (h, torch.from_numpy(case["tok"]).cuda(), torch.from_numpy(case["lp_d"]).cuda(), torch.from_numpy(case["u"]).cuda())
    am1 = torch.empty((B, K), dtype=torch.int32, device="cuda")
    am2 = torch.empty((B, K), dtype=torch.int32, device="cuda")
    a = Kn.LmHeadVerifier(w, B, K)(*args, argmax_out=am1)
    pk = Kn.LmHeadVerifier(w, B, K, packed=True)
    b = pk(*args, argmax_out=am2)
    torch.cuda.synchronize()
    assert torch.equal(a.lp_target, b.lp_target) and torch.equal(a.accept_bits, b.accept_bits) and torch.equal(a.n_acc, b.n_acc)
    assert torch.equal(am1, am2)
    check(dict(lp_t=b.lp_target.cpu().numpy(), accept=b.accept.cpu().numpy(), n_acc=b.n_acc.cpu().numpy(),
               bits=b.accept_bits.cpu().numpy().view(np.uint64)), case["ref"])
    # the shard message of a tensor-parallel head from a packed shard
    m1 = Kn.LmHeadVerifier(w, B, K).partial(h, args[1], 0)
    m2 = pk.partial(h, args[1], 0)
    torch.cuda.synchronize()
    assert torch.equal(m1, m2)


def test_lm_head_verify_strided_operands_and_temperature():
    case = make_case(4, 7, 128, 777, seed=5, ld_h=136, ld_w=200, inv_t=1.0 / 0.7)
    check(run_gpu(case), case["ref"])


def test_lm_head_verify_token_outside_vocabulary_is_rejected():
    case = make_case(2, 4, 64, 200, seed=9)
    case["tok"][0, 1] = -1
    case["tok"][1, 2] = 200
    case["ref"] = O.lm_head_verify(case["hb"], case["wb"], case["tok"], case["lp_d"], case["u"], 2, 4)
    got = run_gpu(case)
    assert got["lp_t"][0, 1] == -np.inf and got["lp_t"][1, 2] == -np.inf
    assert got["accept"][0, 1] == 0 and got["accept"][1, 2] == 0
    check(got, case["ref"])


def test_lm_head_verify_deep_reduction_error_budget():
    """D = 8192 (the 72B lm_head depth) at a reduced vocabulary: the f32 accumulation stays inside LMH_ATOL."""
    case = make_case(8, 8, 8192, 1536, seed=11, scale=4.0)
    got = run_gpu(case)
    err = np.abs(got["lp_t"].astype(np.float64) - case["ref"]["lp_t64"]).max()
    print(f"max |lp - oracle| at D=8192: {err:.3e}")
    check(got, case["ref"])


def make_case_on_gpu(B, K, D, V, seed, scale=3.0):
    """make_case for full-size heads: the operands are drawn on the GPU (1.2 G normals take numpy a minute) and
    their bf16 bit patterns copied back for the f64 oracle, which sees exactly what the kernel reads."""
    import torch

    g = torch.Generator(device="cuda").manual_seed(seed)
    h = torch.randn((B * K, D), device="cuda", generator=g).to(torch.bfloat16)
    w = torch.empty((V, D), dtype=torch.bfloat16, device="cuda")
    for v0 in range(0, V, 16384):
        w[v0:v0 + 16384] = (torch.randn((min(16384, V - v0), D), device="cuda", generator=g) * (scale / D ** 0.5)).to(torch.bfloat16)
    hb = h.view(torch.int16).cpu().numpy().view(np.uint16)
    wb = w.view(torch.int16).cpu().numpy().view(np.uint16)
    rng = np.random.default_rng(seed)
    zeros, half = np.zeros((B, K), np.float32), np.full((B, K), 0.5, np.float32)
    x = O.lm_head_verify(hb, wb, np.zeros((B, K), np.int32), zeros, half, B, K)["logits64"].reshape(B * K, V)
    amax = x.argmax(axis=1)
    tok = np.where(rng.uniform(size=B * K) < 0.7, amax, rng.integers(0, V, B * K)).astype(np.int32).reshape(B, K)
    xs = x.reshape(B, K, V)
    lp_t, _, _ = O.py_verify_accept(xs, tok, zeros, half)
    lp_d = np.minimum(lp_t + rng.normal(0, 0.5, (B, K)), 0.0).astype(np.float32)
    u = rng.uniform(0, 1, (B, K)).astype(np.float32)
    for _ in range(100):
        m = np.abs(np.log(u.astype(np.float64)) - (lp_t - lp_d.astype(np.float64)))
        bad = ~(m >= LMH_MARGIN)
        if not bad.any():
            break
        u[bad] = rng.uniform(0, 1, int(bad.sum())).astype(np.float32)
    lp, acc, n_acc = O.py_verify_accept(xs, tok, lp_d, u)
    bits = np.array([sum(int(f) << k for k, f in enumerate(row)) for row in acc], dtype=np.uint64)
    ref = dict(lp_t64=lp, accept=acc, n_acc=n_acc, bits=bits)
    return dict(B=B, K=K, D=D, V=V, h=h, w=w, tok=tok, lp_d=lp_d, u=u, ref=ref, argmax=amax.reshape(B, K))


@pytest.mark.parametrize("name,D", [("72b", 8192), ("32b", 5120)])
def test_lm_head_verify_real_heads_match_oracle_on_all_rows(name, D):
    """The production shapes (BASELINE configs[2..4]): B = 32, K = 8, V = 152064 with the 72B (D = 8192) and
    14B/32B (D = 5120) lm_head depths -- 512 wide blocks + 82 tail tiles x 3 reduction slices on 256 CUs --
    against the f64 GEMM oracle on all 256 rows, arg-max included."""
    import torch

    from asd_amd import kernels as Kn

    B, K, V = 32, 8, 152064
    case = make_case_on_gpu(B, K, D, V, seed=D)
    ver = Kn.LmHeadVerifier(case["w"], B, K)
    am = torch.empty((B, K), dtype=torch.int32, device="cuda")
    r = ver(case["h"], torch.from_numpy(case["tok"]).cuda(), torch.from_numpy(case["lp_d"]).cuda(),
            torch.from_numpy(case["u"]).cuda(), argmax_out=am)
    torch.cuda.synchronize()
    got = dict(lp_t=r.lp_target.cpu().numpy(), accept=r.accept.cpu().numpy(), n_acc=r.n_acc.cpu().numpy(),
               bits=r.accept_bits.cpu().numpy().view(np.uint64))
    err = np.abs(got["lp_t"].astype(np.float64) - case["ref"]["lp_t64"]).max()
    print(f"{name} head: max |lp - oracle| = {err:.3e}, n_acc = {got['n_acc'].tolist()}")
    check(got, case["ref"])
    # the arg-max may legitimately differ only where the two best f64 logits are closer than the f32 accumulation error
    diff = am.cpu().numpy() != case["argmax"]
    assert diff.sum() <= 1, f"{int(diff.sum())} arg-max rows differ"
    # second call on the same workspace: tickets were re-zeroed by the launcher, results are bit-identical
    r2 = ver(case["h"], torch.from_numpy(case["tok"]).cuda(), torch.from_numpy(case["lp_d"]).cuda(),
             torch.from_numpy(case["u"]).cuda())
    torch.cuda.synchronize()
    assert torch.equal(r2.lp_target, r.lp_target) and torch.equal(r2.accept_bits, r.accept_bits)


def test_lm_head_verify_agrees_with_materialised_logits_at_full_size():
    """BASELINE configs[1] shape with the 7B lm_head (D = 3584, V = 152064): the fused call against
    asd_verify_accept on the f32 logits torch materialises from the same operands."""
    import torch

    from asd_amd import kernels as Kn

    B, K, D, V = 32, 8, 3584, 152064
    g = torch.Generator(device="cuda").manual_seed(3)
    h = torch.randn((B * K, D), device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn((V, D), device="cuda", generator=g) * (3.0 / D ** 0.5)).to(torch.bfloat16)
    logits = (h.float() @ w.float().T).reshape(B, K, V)
    tok = logits.argmax(-1).to(torch.int32)
    tok[:, 1::3] = torch.randint(0, V, tok[:, 1::3].shape, device="cuda", dtype=torch.int32)
    lp_d = -torch.rand((B, K), device="cuda") * 3
    u = torch.rand((B, K), device="cuda")
    ws = Kn.VerifyWorkspace(B, K, V, torch.float32)
    two = Kn.verify_accept(logits, tok, lp_d, u, ws)
    one = Kn.LmHeadVerifier(w, B, K)(h, tok, lp_d, u)
    torch.cuda.synchronize()
    a, b = one.lp_target.double(), two.lp_target.double()
    assert torch.isfinite(b).all()
    assert (a - b).abs().max().item() <= LMH_ATOL
    margin = (torch.log(u.double()) - (b - lp_d.double())).abs()
    safe = margin >= LMH_MARGIN
    assert torch.equal(one.accept[safe], two.accept[safe])
    assert safe.float().mean().item() > 0.9


def test_lm_head_verify_status_codes():
    import torch

    from asd_amd import _binding as Bd

    lib = Bd.load_library()
    B, K, D, V = 2, 3, 64, 100
    h = torch.zeros((B * K, D), dtype=torch.bfloat16, device="cuda")
    w = torch.zeros((V, D), dtype=torch.bfloat16, device="cuda")
    tok = torch.zeros((B, K), dtype=torch.int32, device="cuda")
    f = torch.zeros((B, K), dtype=torch.float32, device="cuda")
    acc = torch.zeros((B, K), dtype=torch.uint8, device="cuda")
    n = torch.zeros((B,), dtype=torch.int32, device="cuda")
    nbytes = lib.asd_lm_head_verify_workspace_bytes(B, K, V)
    assert nbytes >= 1 * B * K * 12 and nbytes % 256 == 0
    ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")

    def call(**kw):
        a = dict(h=h.data_ptr(), ld_h=D, w=w.data_ptr(), ld_w=D, dtype=Bd.DTYPE_BF16, D=D, B=B, K=K, V=V, inv_t=1.0,
                 ws=ws.data_ptr(), ws_bytes=nbytes)
        a.update(kw)
        return lib.asd_lm_head_verify(a["h"], a["ld_h"], a["w"], a["ld_w"], a["dtype"], a["D"], tok.data_ptr(),
                                      f.data_ptr(), f.data_ptr(), a["B"], a["K"], a["V"], a["inv_t"], f.data_ptr(),
                                      acc.data_ptr(), n.data_ptr(), None, a["ws"], a["ws_bytes"], None)

    assert call() == 0
    assert call(B=0) == 0
    assert call(dtype=Bd.DTYPE_F16) == 0           # f16 operands are served (the same bytes read as f16)
    assert call(dtype=Bd.DTYPE_F32) == -2
    assert call(D=48, ld_h=48, ld_w=48) == -2
    assert call(D=96, ld_h=96, ld_w=96) == -2      # whole 64-column superstages only
    assert call(K=65) == -2
    assert call(ld_h=D - 8) == -1
    assert call(inv_t=0.0) == -1
    assert call(h=None) == -1
    assert call(ld_w=D + 4) == -5
    assert call(h=h.data_ptr() + 2) == -5
    assert call(ws_bytes=nbytes - 256) == -3
    torch.cuda.synchronize()


def test_model_tier_hands_over_hidden_states():
    """SyntheticLM(return_hidden) + SpeculativeVerifier.step_hidden against the f64 oracle on the same
    hidden states / lm_head matrix, and against the logits path (which rounds the logits to bf16 first:
    agreement there is to that rounding only)."""
    import torch

    from asd_amd.minimal_adaptive_decoder import MinimalQualityPredictor
    from asd_amd.serving import synthetic_lm as SL
    from asd_amd.serving.speculative import SpeculativeVerifier

    B, K, V = 4, 6, 1500
    lm = SL.SyntheticLM(SL.tiny(vocab=V, hidden=128), device="cuda", seed=3, logit_scale=2.0)
    g = torch.Generator(device="cuda").manual_seed(0)
    ids = torch.randint(0, V, (B, K), device="cuda", generator=g)
    hidden = lm(ids, return_hidden=True)
    lm.reset()
    logits = lm(ids)
    tok = logits.argmax(-1).to(torch.int32)
    tok[:, 1::2] = torch.randint(0, V, tok[:, 1::2].shape, device="cuda", generator=g, dtype=torch.int32)
    lp_d = -torch.rand((B, K), device="cuda", generator=g) * 2
    u = torch.rand((B, K), device="cuda", generator=g)
    torch.manual_seed(0)
    ver = SpeculativeVerifier(B, K, V, predictor=MinimalQualityPredictor())
    ver.inv_temperature = 1.0 / 0.7
    feat = torch.randn((B, 64), device="cuda", generator=g)
    res = ver.step_hidden(hidden, lm.lm_head.weight, tok, lp_d, u, feat, logit_scale=lm.logit_scale)
    two = ver.step(logits, tok, lp_d, u, feat)
    torch.cuda.synchronize()
    hb = hidden.reshape(B * K, -1).view(torch.int16).cpu().numpy().view(np.uint16)
    wb = lm.lm_head.weight.detach().view(torch.int16).cpu().numpy().view(np.uint16)
    inv_t = float(np.float32(np.float32(1.0 / 0.7) * np.float32(2.0)))
    ref = O.lm_head_verify(hb, wb, tok.cpu().numpy(), lp_d.cpu().numpy(), u.cpu().numpy(), B, K, inv_t)
    np.testing.assert_allclose(res.verify.lp_target.cpu().numpy(), ref["lp_t64"], rtol=0, atol=LMH_ATOL)
    safe = ref["margin"] >= LMH_MARGIN
    assert np.array_equal(res.verify.accept.cpu().numpy()[safe], ref["accept"][safe])
    assert res.stop is not None and res.stop.score.shape == (B,)
    assert (res.verify.lp_target - two.verify.lp_target).abs().max().item() < 0.1     # bf16 rounding of the logits


def test_token_logprobs_from_hidden():
    import torch

    from asd_amd.training.logprobs import token_logprobs, token_logprobs_from_hidden

    T, D, V = 70, 128, 900                   # T > 64: two launches
    g = torch.Generator(device="cuda").manual_seed(5)
    h = torch.randn((T, D), device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn((V, D), device="cuda", generator=g) * (2.0 / D ** 0.5)).to(torch.bfloat16)
    tok = torch.randint(0, V, (T,), device="cuda", generator=g)
    got = token_logprobs_from_hidden(h, w, tok).cpu().numpy()
    exact = torch.log_softmax(h.double() @ w.double().T, -1).gather(1, tok[:, None])[:, 0].cpu().numpy()
    np.testing.assert_allclose(got, exact, rtol=0, atol=LMH_ATOL)
    scored = token_logprobs((h.float() @ w.float().T), tok)       # the reference-shaped call on materialised f32 scores
    np.testing.assert_allclose(got, scored, rtol=0, atol=LMH_ATOL)


@pytest.mark.parametrize("B,K,D,V", [(3, 5, 64, 300), (32, 8, 256, 4173), (2, 7, 128, 66000)])
def test_lm_head_argmax_and_greedy_verification(B, K, D, V):
    """asd_lm_head_verify_ex: the row arg-max and greedy accept (tok == argmax).  The arg-max is compared
    where the f64 top-2 gap exceeds the f32 accumulation error (ties / near-ties are a coin flip for any
    finite-precision GEMM); on those rows it must be exact, and the greedy mask follows from it."""
    import torch

    from asd_amd import kernels as Kn

    case = make_case(B, K, D, V, seed=7 * B + K)
    x = case["ref"]["logits64"].reshape(B * K, V)
    top2 = np.partition(x, V - 2, axis=1)[:, V - 2:]
    clear = (top2[:, 1] - top2[:, 0]) > 1e-3
    assert clear.mean() > 0.9
    am_ref = x.argmax(axis=1).astype(np.int32)
    tok = case["tok"].copy().reshape(-1)
    tok[::3] = am_ref[::3]                                   # make a third of the draft tokens the arg-max
    tok = tok.reshape(B, K)
    w, h = bf16_dev(case["wb"]), bf16_dev(case["hb"])
    ver = Kn.LmHeadVerifier(w, B, K)
    am = torch.full((B, K), -7, dtype=torch.int32, device="cuda")
    r = ver(h, torch.from_numpy(tok).cuda(), greedy=True, argmax_out=am)
    torch.cuda.synchronize()
    am_gpu = am.cpu().numpy().reshape(-1)
    assert np.array_equal(am_gpu[clear], am_ref[clear])
    acc = r.accept.cpu().numpy().reshape(-1)
    assert np.array_equal(acc, (tok.reshape(-1) == am_gpu).astype(np.uint8))      # the mask is the kernel's own arg-max test
    n_acc_ref = np.array([int(np.argmin(np.append(a, 0))) for a in acc.reshape(B, K)], dtype=np.int32)
    assert np.array_equal(r.n_acc.cpu().numpy(), n_acc_ref)
    ref = O.lm_head_verify(case["hb"], case["wb"], tok, np.zeros((B, K), np.float32), np.ones((B, K), np.float32), B, K)
    np.testing.assert_allclose(r.lp_target.cpu().numpy(), ref["lp_t64"], rtol=0, atol=LMH_ATOL)   # lp is still reported
    # the sampling call reports the same arg-max
    am2 = torch.empty_like(am)
    ver(h, torch.from_numpy(tok).cuda(), torch.from_numpy(case["lp_d"]).cuda(), torch.from_numpy(case["u"]).cuda(), argmax_out=am2)
    torch.cuda.synchronize()
    assert torch.equal(am, am2)


def test_greedy_loop_without_target_logits_reproduces_target_greedy_decoding():
    """N2 + N3 end to end: greedy speculative decoding where the target tier only ever hands over hidden
    states.  Teacher-forced check: at every generated position the committed token is the arg-max of the
    target's f32 logits given the committed prefix (positions whose top-2 gap is below 1e-2 are skipped)."""
    import torch

    from asd_amd.serving import synthetic_lm as SL
    from asd_amd.serving.speculative import SpeculativeVerifier, speculative_generate_ragged

    B, K, V, P, NEW = 4, 4, 1000, 5, 20
    target = SL.SyntheticLM(SL.tiny(vocab=V, hidden=128, layers=2), device="cuda", seed=1, logit_scale=4.0)
    draft = SL.SyntheticLM(SL.tiny(vocab=V, hidden=128, layers=2), device="cuda", seed=1, logit_scale=4.0)
    with torch.no_grad():                                   # a draft that agrees with the target most of the time
        for pd in draft.parameters():
            pd.add_(torch.randn_like(pd) * 0.01 * pd.abs().mean())
    g = torch.Generator(device="cuda").manual_seed(2)
    prompt = torch.randint(0, V, (B, P), device="cuda", generator=g)
    tr = speculative_generate_ragged(draft, target, prompt, NEW, SpeculativeVerifier(B, K, V), greedy_hidden=True,
                                     keep_inputs=True, sync_every=2)
    assert (tr.seq_len.cpu().numpy() == P + NEW).all()
    toks = tr.tokens.to(torch.int64)
    target.reset()
    hid = target(toks[:, :-1], return_hidden=True)                         # dense forward over the committed text
    logits = hid.float() @ target.lm_head.weight.float().T
    top2 = logits.topk(2, dim=-1).values
    clear = (top2[..., 0] - top2[..., 1]) > 1e-2
    pred = logits.argmax(-1)                                               # pred[:, t] = greedy token at position t + 1
    gen_pos = torch.arange(P - 1, P + NEW - 1, device="cuda")
    ok = clear[:, gen_pos]
    assert ok.float().mean().item() > 0.8
    assert torch.equal(pred[:, gen_pos][ok], toks[:, P:][ok])
    assert tr.steps < NEW                                                  # the draft got tokens accepted
    assert sum(int(m.sum().item()) for m in tr.accept_masks) > 0


def test_lm_head_partial_shards_combine_to_the_full_call():
    """asd_lm_head_partial: two vocabulary shards (ragged split) reduced separately, stacked like an
    all-gather and finished with asd_accept_from_partials, against the oracle and the unsharded call."""
    import torch

    from asd_amd import kernels as Kn
    from asd_amd.distributed import shard_bounds

    B, K, D, V = 6, 8, 128, 5001
    case = make_case(B, K, D, V, seed=21, inv_t=1.0 / 0.8)
    w, h = bf16_dev(case["wb"]), bf16_dev(case["hb"])
    tok = torch.from_numpy(case["tok"]).cuda()
    lp_d, u = torch.from_numpy(case["lp_d"]).cuda(), torch.from_numpy(case["u"]).cuda()
    msgs = []
    for rank in range(3):
        v0, v1 = shard_bounds(V, 3, rank)
        shard = w[v0:v1]                                    # a view: rows stay 16-byte aligned (D = 128)
        msgs.append(Kn.LmHeadVerifier(shard, B, K).partial(h, tok, v0, inv_temperature=case["inv_t"]))
    r = Kn.accept_from_partials(torch.stack(msgs).contiguous(), lp_d, u, inv_temperature=case["inv_t"])
    full = Kn.LmHeadVerifier(w, B, K)(h, tok, lp_d, u, inv_temperature=case["inv_t"])
    torch.cuda.synchronize()
    got = dict(lp_t=r.lp_target.cpu().numpy(), accept=r.accept.cpu().numpy(), n_acc=r.n_acc.cpu().numpy(),
               bits=r.accept_bits.cpu().numpy().view(np.uint64))
    check(got, case["ref"])
    assert (r.lp_target - full.lp_target).abs().max().item() < 1e-5
    assert torch.equal(r.accept, full.accept)


def test_lm_head_verify_non_finite_rows():
    """A NaN hidden row makes every logit of that row NaN: lp_t is NaN, the row is rejected and has no
    arg-max (-1); an all-zero hidden row gives the uniform distribution (lp = -log V) and arg-max 0
    (ties -> lowest id).  The other rows are unaffected."""
    import torch

    from asd_amd import kernels as Kn

    B, K, D, V = 2, 4, 64, 333
    case = make_case(B, K, D, V, seed=31)
    h = bf16_dev(case["hb"]).clone()
    h[1] = float("nan")                      # row (0, 1)
    h[6] = 0.0                               # row (1, 2)
    w = bf16_dev(case["wb"])
    tok = torch.from_numpy(case["tok"]).cuda()
    am = torch.empty((B, K), dtype=torch.int32, device="cuda")
    r = Kn.LmHeadVerifier(w, B, K)(h, tok, torch.from_numpy(case["lp_d"]).cuda(), torch.from_numpy(case["u"]).cuda(),
                                   argmax_out=am)
    torch.cuda.synchronize()
    lp = r.lp_target.cpu().numpy()
    assert np.isnan(lp[0, 1]) and r.accept[0, 1].item() == 0 and am[0, 1].item() == -1
    assert abs(lp[1, 2] + np.log(V)) < 1e-5 and am[1, 2].item() == 0
    keep = np.ones((B, K), bool)
    keep[0, 1] = keep[1, 2] = False
    np.testing.assert_allclose(lp[keep], case["ref"]["lp_t64"][keep], rtol=0, atol=LMH_ATOL)
    assert r.n_acc[0].item() <= 1            # the NaN row ends sequence 0's accepted prefix


def test_lm_head_verify_graph_replay_and_reproducibility():
    """The fused call (incl. the sliced tail: a memset node + three kernels) captured in a hipGraph and replayed
    on changing inputs equals the eager call bit for bit, and repeated eager calls are bit-identical (the
    reduction slices are summed in slice order whatever their arrival order)."""
    import torch

    from asd_amd import kernels as Kn

    B, K, D, V = 32, 8, 4608, 65700          # 256 wide blocks + one ragged 256-column block in 3 reduction slices
    g = torch.Generator(device="cuda").manual_seed(9)
    w = (torch.randn((V, D), device="cuda", generator=g) * (3.0 / D ** 0.5)).to(torch.bfloat16)
    h = torch.randn((B * K, D), device="cuda", generator=g).to(torch.bfloat16)
    tok = torch.randint(0, V, (B, K), device="cuda", generator=g, dtype=torch.int32)
    lp_d = -torch.rand((B, K), device="cuda", generator=g)
    u = torch.rand((B, K), device="cuda", generator=g)
    ver = Kn.LmHeadVerifier(w, B, K)
    first = ver(h, tok, lp_d, u)
    ref_lp = first.lp_target.clone()
    for _ in range(5):
        again = ver(h, tok, lp_d, u)
        assert torch.equal(again.lp_target, ref_lp)
    out = ver(h, tok, lp_d, u)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        ver(h, tok, lp_d, u, out=out)                      # warm-up on the capture stream
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        ver(h, tok, lp_d, u, out=out)
    for seed in (1, 2, 3):
        g2 = torch.Generator(device="cuda").manual_seed(seed)
        h.copy_(torch.randn((B * K, D), device="cuda", generator=g2).to(torch.bfloat16))
        u.copy_(torch.rand((B, K), device="cuda", generator=g2))
        graph.replay()
        torch.cuda.synchronize()
        got_lp, got_acc = out.lp_target.clone(), out.accept.clone()
        eager = ver(h, tok, lp_d, u)
        torch.cuda.synchronize()
        assert torch.equal(got_lp, eager.lp_target) and torch.equal(got_acc, eager.accept)
