"""X3: the decoder-layer kernels (csrc/decoder.hip) and the HipDecoder stack against plain PyTorch fp32 references of the
same ops on the same bf16 inputs.  Floating point: tolerances are stated per test (one bf16 rounding of the result = 2^-9
relative, plus the arithmetic differences named there)."""
import math
from importlib import import_module

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

BF = torch.bfloat16


def _lib():
    return import_module("adaptive-speculative-decoding_amd.kernels")._lib()


def _B():
    return import_module("adaptive-speculative-decoding_amd._binding")


def _check(rc, name):
    _B().check(name, rc)


def _gen(seed):
    return torch.Generator(device="cuda").manual_seed(seed)


@pytest.mark.parametrize("M,D", [(1, 128), (32, 3584), (288, 5120), (7, 8192), (3, 1000)])
def test_rmsnorm(M, D):
    g = _gen(M + D)
    x = (torch.randn(M, D, generator=g, device="cuda") * 3).to(BF)
    w = (1 + 0.1 * torch.randn(D, generator=g, device="cuda")).to(BF)
    out = torch.empty_like(x)
    _check(_lib().asd_rmsnorm(x.data_ptr(), D, w.data_ptr(), 1e-6, _B().DTYPE_BF16, M, D, out.data_ptr(), D, None), "asd_rmsnorm")
    v = x.double()
    ref = v * torch.rsqrt(v.pow(2).mean(-1, keepdim=True) + 1e-6) * w.double()
    err = (out.double() - ref).abs()
    assert bool((err <= 2.0 ** -8 * ref.abs() + 1e-6).all())          # f32 arithmetic + one bf16 rounding


@pytest.mark.parametrize("M,I", [(1, 64), (32, 18944), (288, 27648)])
def test_silu_mul(M, I):
    g = _gen(M + I)
    gu = (torch.randn(M, 2 * I, generator=g, device="cuda") * 2).to(BF)
    act = torch.empty(M, I, dtype=BF, device="cuda")
    _check(_lib().asd_silu_mul(gu.data_ptr(), 2 * I, _B().DTYPE_BF16, M, I, act.data_ptr(), I, None), "asd_silu_mul")
    ref = F.silu(gu[:, :I].double()) * gu[:, I:].double()
    err = (act.double() - ref).abs()
    assert bool((err <= 2.0 ** -8 * ref.abs() + 1e-6).all())


def _rope_ref(x, pos, theta):
    """x [M, heads, 128] f64, pos [M]"""
    d = x.shape[-1]
    inv = 1.0 / (theta ** (torch.arange(0, d, 2, device=x.device, dtype=torch.float32) / d))
    ang = (pos.float()[:, None] * inv).double()[:, None, :]
    x1, x2 = x[..., : d // 2], x[..., d // 2:]
    return torch.cat([x1 * ang.cos() - x2 * ang.sin(), x2 * ang.cos() + x1 * ang.sin()], dim=-1)


def _fill_cache(Bc, KVH, t_max, seed):
    g = _gen(seed)
    k = torch.randn(Bc, KVH, t_max, 128, generator=g, device="cuda").to(BF)
    v = torch.randn(Bc, KVH, t_max, 128, generator=g, device="cuda").to(BF)
    return k, v


@pytest.mark.parametrize("Bn,T,H,KVH,subset", [(4, 1, 28, 4, False), (3, 9, 40, 8, True), (2, 9, 64, 8, False)])
def test_rope_kv_store(Bn, T, H, KVH, subset):
    t_max, theta = 64, 1.0e6
    g = _gen(Bn * 100 + T)
    M = Bn * T
    width = (H + 2 * KVH) * 128
    qkv = torch.randn(M, width, generator=g, device="cuda").to(BF)
    pos0 = torch.randint(0, t_max - T, (Bn,), generator=g, device="cuda")
    pos0[-1] = t_max - 1                                            # its later positions fall off the cache: clamped to t_max - 1
    pos_raw = (pos0[:, None] + torch.arange(T, device="cuda")).reshape(M).to(torch.int32)
    pos = pos_raw.clamp(max=t_max - 1)
    Bc = Bn + 2
    rows = torch.tensor([Bc - 1 - i for i in range(Bn)], dtype=torch.int32, device="cuda") if subset else None
    kc = torch.zeros(Bc, KVH, t_max, 128, dtype=BF, device="cuda")
    vt = torch.zeros(Bc, KVH, 128, t_max, dtype=BF, device="cuda")
    inv = (1.0 / (theta ** (torch.arange(0, 128, 2, device="cuda", dtype=torch.float32) / 128))).contiguous()
    orig = qkv.clone()
    _check(_lib().asd_rope_kv_store(qkv.data_ptr(), width, pos_raw.data_ptr(), None if rows is None else rows.data_ptr(), inv.data_ptr(),
                                    _B().DTYPE_BF16, Bn, T, H, KVH, 128, kc.data_ptr(), vt.data_ptr(), t_max, None), "asd_rope_kv_store")
    o = orig.double().view(M, H + 2 * KVH, 128)
    q_ref = _rope_ref(o[:, :H], pos, theta)
    k_ref = _rope_ref(o[:, H:H + KVH], pos, theta)
    got_q = qkv.double().view(M, H + 2 * KVH, 128)[:, :H]
    # angle = f32(pos) * f32 inv_freq in both; cosf / sinf vs f64 cos / sin of the same f32 angle: <= 1e-6; one bf16 rounding
    assert bool(((got_q - q_ref).abs() <= 2.0 ** -8 * q_ref.abs() + 1e-5).all())
    assert torch.equal(qkv.view(M, H + 2 * KVH, 128)[:, H:], orig.view(M, H + 2 * KVH, 128)[:, H:])      # k, v columns untouched
    for m in range(M):
        b = m // T
        row = int(rows[b]) if subset else b
        p = int(pos[m])
        if T > 1 and b == Bn - 1:
            continue                       # several positions of this feed share the last slot: which one it keeps is not defined
        assert bool(((kc[row, :, p].double() - k_ref[m]).abs() <= 2.0 ** -8 * k_ref[m].abs() + 1e-5).all())
        assert torch.equal(vt[row, :, :, p], orig.view(M, H + 2 * KVH, 128)[m, H + KVH:])
    written = torch.zeros(Bc, t_max, dtype=torch.bool, device="cuda")
    for m in range(M):
        written[int(rows[m // T]) if subset else m // T, int(pos[m])] = True
    assert bool((kc.abs().sum((1, 3)) == 0)[~written].all()) and bool((vt.abs().sum((1, 2)) == 0)[~written].all())


@pytest.mark.parametrize("Bn,T,H,KVH,t_max,subset", [
    (32, 1, 28, 4, 256, False), (5, 9, 28, 4, 128, True), (4, 9, 40, 8, 96, False), (3, 9, 64, 8, 160, True), (2, 3, 8, 8, 32, False),
    (2, 40, 16, 2, 320, False),
])
def test_attn_ragged_matches_fp32_softmax(Bn, T, H, KVH, t_max, subset):
    g = _gen(Bn * 1000 + T * 10 + H)
    M = Bn * T
    Bc = Bn + 1
    kc, v = _fill_cache(Bc, KVH, t_max, seed=Bn + T)
    vt = v.transpose(2, 3).contiguous()
    width = (H + 2 * KVH) * 128
    qkv = torch.randn(M, width, generator=g, device="cuda").to(BF)
    pos0 = torch.randint(0, t_max - T + 1, (Bn,), generator=g, device="cuda")
    pos0[0] = 0                                                     # a sequence at its very first token
    pos0[-1] = t_max - T                                            # ... and one that fills the cache
    pos = (pos0[:, None] + torch.arange(T, device="cuda")).reshape(M).to(torch.int32)
    rows = torch.tensor([Bc - 1 - i for i in range(Bn)], dtype=torch.int32, device="cuda") if subset else None
    out = torch.full((M, H * 128), 7.0, dtype=BF, device="cuda")
    _check(_lib().asd_attn_ragged(qkv.data_ptr(), width, kc.data_ptr(), vt.data_ptr(), pos.data_ptr(),
                                  None if rows is None else rows.data_ptr(), _B().DTYPE_BF16, Bn, T, H, KVH, 128, t_max,
                                  out.data_ptr(), H * 128, None), "asd_attn_ragged")
    rep = H // KVH
    q = qkv.double().view(M, H + 2 * KVH, 128)[:, :H]
    worst = 0.0
    for m in range(M):
        b = m // T
        row = int(rows[b]) if subset else b
        L = int(pos[m]) + 1
        kk = kc[row, :, :L].double().repeat_interleave(rep, dim=0)          # [H, L, 128]
        vv = v[row, :, :L].double().repeat_interleave(rep, dim=0)
        sc = torch.einsum("hd,hld->hl", q[m], kk) / math.sqrt(128.0)
        ref = torch.einsum("hl,hld->hd", torch.softmax(sc, dim=-1), vv)
        got = out[m].double().view(H, 128)
        # P is rounded to bf16 before the second product (2^-9 per weight, as in flash attention), exp2 approximations
        # (~1 ulp f32), one bf16 rounding of the output
        worst = max(worst, float((got - ref).abs().max()))
        assert bool(((got - ref).abs() <= 2.0 ** -7 * ref.abs() + 6e-3).all()), (m, float((got - ref).abs().max()))
    assert worst > 0.0


def _small_shape():
    SL = import_module("adaptive-speculative-decoding_amd.serving.synthetic_lm")
    return SL.LMShape("small", 512, 3, 4, 2, 1024, vocab=1000, rope_theta=1.0e6)


def _run_sequence(lm, Bn, prompt, steps, T, seed):
    """prefill, `steps` feeds of T tokens with a rollback in between; returns the list of logits"""
    g = _gen(seed)
    lm.alloc_ragged(Bn, 96)
    outs = []
    ids = torch.randint(0, lm.shape.vocab, (Bn, prompt), generator=g, device="cuda")
    length = torch.full((Bn,), 0, dtype=torch.int64, device="cuda")
    outs.append(lm.forward_ragged(ids, length, prompt))
    length += prompt
    for i in range(steps):
        ids = torch.randint(0, lm.shape.vocab, (Bn, T), generator=g, device="cuda")
        outs.append(lm.forward_ragged(ids, length, int(length.max()) + T))
        keep = torch.randint(1, T + 1, (Bn,), generator=g, device="cuda")          # commit a prefix: ragged lengths, KV rollback
        length += keep
    sub = torch.tensor([Bn - 1, 0], device="cuda")                                    # a subset feed (tier escalation)
    ids = torch.randint(0, lm.shape.vocab, (2, T), generator=g, device="cuda")
    outs.append(lm.forward_ragged(ids, length[sub], int(length.max()) + T, rows=sub))
    return outs


def test_hip_decoder_matches_torch_modules_and_fp32():
    SL = import_module("adaptive-speculative-decoding_amd.serving.synthetic_lm")
    shape = _small_shape()
    ref32 = SL.SyntheticLM(shape, dtype=BF, device="cuda", seed=4).float()            # the same bf16 weights, fp32 arithmetic
    lm_t = SL.SyntheticLM(shape, dtype=BF, device="cuda", seed=4)
    lm_h = SL.SyntheticLM(shape, dtype=BF, device="cuda", seed=4)
    lm_h.enable_hip_layers()
    for a, b in zip(lm_t.parameters(), lm_h.parameters()):
        assert torch.equal(a, b)                                                     # fusing q|k|v and gate|up moved no value
    o32 = _run_sequence(ref32, 3, 20, 4, 5, seed=8)
    ot = _run_sequence(lm_t, 3, 20, 4, 5, seed=8)
    oh = _run_sequence(lm_h, 3, 20, 4, 5, seed=8)
    for r, t, h in zip(o32, ot, oh):
        assert h.shape == t.shape and h.dtype == BF
        et = (t.float() - r).abs().max().item()
        eh = (h.float() - r).abs().max().item()
        scale = r.abs().max().item()
        # both bf16 pipelines sit within bf16 noise of the fp32 model; the HIP stack rounds less often than the modules
        assert eh <= max(2.0 * et, 0.02 * scale), (eh, et, scale)


def test_decoder_forward_equals_the_composition_of_its_parts():
    """asd_decoder_forward lets the kernel after a sliced projection add the f32 partials itself; the exported parts, called
    one by one with asd_linear_ex's own reduce kernel in between, must give the same bits."""
    SL = import_module("adaptive-speculative-decoding_amd.serving.synthetic_lm")
    K_ = import_module("adaptive-speculative-decoding_amd.kernels")
    lib, Bd = _lib(), _B()
    shape = _small_shape()
    lm = SL.SyntheticLM(shape, dtype=BF, device="cuda", seed=6)
    lm.enable_hip_layers()
    Bn, T, t_max = 3, 5, 64
    lm.alloc_ragged(Bn, t_max)
    hd = lm._hip
    g = _gen(2)
    ids = torch.randint(0, shape.vocab, (Bn, T), generator=g, device="cuda")
    pos0 = torch.tensor([0, 7, 30], device="cuda")
    # the composition, on its own caches
    M = Bn * T
    x = lm.embed(ids).view(M, shape.hidden).clone()
    pos = (pos0[:, None] + torch.arange(T, device="cuda")).reshape(M).to(torch.int32)
    ws = K_.LinearWorkspace("cuda")
    kv = shape.kv_heads * 128
    for i, blk in enumerate(lm.blocks):
        qkv_w, qkv_b, gu_w = hd._fused[i]
        kc, vt = torch.zeros_like(hd.k_cache[i]), torch.zeros_like(hd.vt_cache[i])
        hn = torch.empty_like(x)
        _check(lib.asd_rmsnorm(x.data_ptr(), shape.hidden, blk.ln1.weight.data_ptr(), shape.rms_eps, Bd.DTYPE_BF16, M, shape.hidden, hn.data_ptr(), shape.hidden, None), "asd_rmsnorm")
        qkv = K_.linear(hn, qkv_w, qkv_b, workspace=ws)
        _check(lib.asd_rope_kv_store(qkv.data_ptr(), qkv.stride(0), pos.data_ptr(), None, hd.inv_freq.data_ptr(), Bd.DTYPE_BF16, Bn, T,
                                     shape.heads, shape.kv_heads, 128, kc.data_ptr(), vt.data_ptr(), hd.t_max, None), "asd_rope_kv_store")
        attn = torch.empty(M, shape.hidden, dtype=BF, device="cuda")
        _check(lib.asd_attn_ragged(qkv.data_ptr(), qkv.stride(0), kc.data_ptr(), vt.data_ptr(), pos.data_ptr(), None, Bd.DTYPE_BF16, Bn, T,
                                   shape.heads, shape.kv_heads, 128, hd.t_max, attn.data_ptr(), shape.hidden, None), "asd_attn_ragged")
        K_.linear(attn, blk.o.weight, workspace=ws, out=x, residual=x)
        _check(lib.asd_rmsnorm(x.data_ptr(), shape.hidden, blk.ln2.weight.data_ptr(), shape.rms_eps, Bd.DTYPE_BF16, M, shape.hidden, hn.data_ptr(), shape.hidden, None), "asd_rmsnorm")
        gu = K_.linear(hn, gu_w, workspace=ws)
        act = torch.empty(M, shape.intermediate, dtype=BF, device="cuda")
        _check(lib.asd_silu_mul(gu.data_ptr(), gu.stride(0), Bd.DTYPE_BF16, M, shape.intermediate, act.data_ptr(), shape.intermediate, None), "asd_silu_mul")
        K_.linear(act, blk.down.weight, workspace=ws, out=x, residual=x)
    want = torch.empty_like(x)
    _check(lib.asd_rmsnorm(x.data_ptr(), shape.hidden, lm.norm.weight.data_ptr(), shape.rms_eps, Bd.DTYPE_BF16, M, shape.hidden, want.data_ptr(), shape.hidden, None), "asd_rmsnorm")
    got = lm.forward_ragged(ids, pos0, 40, return_hidden=True)
    assert lib.asd_linear_slices(M, shape.hidden, shape.hidden) > 1           # the fused consumers really ran
    assert torch.equal(got.view(M, shape.hidden), want)


def test_hip_decoder_graph_replay_equals_eager():
    SL = import_module("adaptive-speculative-decoding_amd.serving.synthetic_lm")
    shape = _small_shape()
    a = SL.SyntheticLM(shape, dtype=BF, device="cuda", seed=5)
    b = SL.SyntheticLM(shape, dtype=BF, device="cuda", seed=5)
    a.enable_hip_layers()
    b.enable_hip_layers()
    b.enable_graphs(True)
    g = _gen(1)
    for lm in (a, b):
        lm.alloc_ragged(4, 64)
    ids = torch.randint(0, 1000, (4, 8), generator=g, device="cuda")
    z = torch.zeros(4, dtype=torch.int64, device="cuda")
    ra, rb = a.forward_ragged(ids, z, 8), b.forward_ragged(ids, z, 8)
    assert torch.equal(ra, rb)
    for step in range(3):
        ids = torch.randint(0, 1000, (4, 1), generator=g, device="cuda")
        ra, rb = a.forward_ragged(ids, z + 8 + step, 9 + step), b.forward_ragged(ids, z + 8 + step, 9 + step).clone()
        assert torch.equal(ra, rb)


def test_hip_decoder_refuses_cpu_and_other_head_dims():
    SL = import_module("adaptive-speculative-decoding_amd.serving.synthetic_lm")
    with pytest.raises(RuntimeError):
        SL.SyntheticLM(SL.tiny(), device="cpu").enable_hip_layers()
    with pytest.raises(RuntimeError):
        SL.SyntheticLM(SL.tiny(), device="cuda").enable_hip_layers()                  # head_dim 32


def test_three_tier_loop_with_hip_decoder_models():
    """The stop-or-escalate loop with every tier's passes going through asd_decoder_forward: accept masks against the oracle on
    the recorded logits (bit-exact outside the stated margin), committed lengths, and -- the KV-rollback check -- the logits of
    the stateful model against a from-scratch pass of a fresh model over each sequence's committed context."""
    import numpy as np
    from oracle import oracle as O
    from asd_amd.distributed import HipOps
    from asd_amd.serving import hierarchy as H
    from tests.test_hierarchy import _predictor
    SL = import_module("adaptive-speculative-decoding_amd.serving.synthetic_lm")
    V, Bn, P, NEW, Kd = 1000, 6, 5, 24, 4
    shape = SL.LMShape("small", 512, 2, 4, 2, 1024, vocab=V, rope_theta=1.0e6)
    prompt = torch.randint(0, V, (Bn, P), generator=torch.Generator().manual_seed(7)).cuda()
    cfg = H.HierarchyConfig(draft_len=Kd, temperature=0.7, top_p=0.9, lambda_value=25.0, seed=3)
    pl = H.Placement.for_world(1)
    kw = dict(heads=("logits", "logits"), logit_scale=4.0, keep_inputs=True, weight_noise=(0.0, 0.02, 0.04), share_seed=1)
    draft, tiers = H.build_rank_roles(0, pl, [shape] * 3, cfg, prompt, NEW, _predictor(), ops=HipOps(), hip_layers=True, **kw)
    assert draft.m.execution == "hip_decoder" and all(t.m.execution == "hip_decoder" for t in tiers.values())
    tr = H.generate_hierarchical(draft, [tiers[1], tiers[2]], keep_inputs=True)
    assert (tr.seq_len == P + NEW).all()
    assert tr.tier_calls[1] > 0
    fresh = SL.SyntheticLM(shape, dtype=BF, device="cuda", seed=1, logit_scale=4.0)
    fresh.lm_head.weight.data.copy_(tiers[1].m.lm_head.weight.data)
    fresh.enable_hip_layers()
    inv_t = np.float32(1 / 0.7)
    lens = np.full(Bn, P)
    buf = np.zeros((Bn, P + NEW), np.int32)
    buf[:, :P] = prompt.cpu().numpy()
    checked, compared = 0, 0
    for rec in tr.records:
        dm, final = rec["draft"], rec["final"]
        if 1 in rec["tiers"]:
            v, _ = rec["tiers"][1]
            idx = v.idx.cpu().numpy()
            inp, n = v.inputs, len(idx)
            tok, lp_d, u = (inp[k].cpu().numpy() for k in ("tok", "lp_d", "u"))
            store = inp["logits"].contiguous().view(torch.int16).cpu().numpy().view(np.uint16)
            ref = O.verify_accept(store.reshape(n * Kd, V), O.DT_BF16, tok, lp_d, u, n, Kd, V, inv_temperature=inv_t)
            safe = ref["margin"] >= 1e-4
            assert np.array_equal(v.accept.cpu().numpy()[safe], ref["accept"][safe])
            checked += int(safe.sum())
            for i, b in enumerate(idx[:2]):                      # from-scratch pass over the committed context + the draft
                ctx = np.concatenate([buf[b, :lens[b]], tok[i]])
                fresh.alloc_ragged(1, 64)
                full = fresh.forward_ragged(torch.from_numpy(ctx[None, :]).to(torch.int64).cuda(), torch.zeros(1, dtype=torch.int64, device="cuda"), len(ctx))[0]
                got = inp["logits"][i].float()
                want = full[lens[b] - 1: lens[b] - 1 + Kd].float()
                # same weights, same kernels, but other row counts per call (other slice plans, other MFMA tilings): bf16 noise
                assert (got - want).abs().max().item() <= 0.05 * want.abs().max().item()
                compared += 1
        tokc = dm.tok.cpu().numpy()
        for b in range(Bn):
            new = (list(tokc[b, :int(final.n_acc[b])]) + [int(final.drawn[b])])[: max(0, P + NEW - lens[b])]
            buf[b, lens[b]:lens[b] + len(new)] = new
            lens[b] += len(new)
    assert np.array_equal(buf, tr.tokens.cpu().numpy())
    assert checked > 50 and compared > 4


def test_decoder_entry_points_reject_bad_arguments():
    lib, Bd = _lib(), _B()
    x = torch.zeros(4, 256, dtype=BF, device="cuda")
    w = torch.ones(256, dtype=BF, device="cuda")
    out = torch.empty_like(x)
    ok = lib.asd_rmsnorm(x.data_ptr(), 256, w.data_ptr(), 1e-6, Bd.DTYPE_BF16, 4, 256, out.data_ptr(), 256, None)
    assert ok == 0
    assert lib.asd_rmsnorm(x.data_ptr(), 256, w.data_ptr(), 1e-6, Bd.DTYPE_F16, 4, 256, out.data_ptr(), 256, None) < 0       # bf16 only
    assert lib.asd_rmsnorm(x.data_ptr(), 128, w.data_ptr(), 1e-6, Bd.DTYPE_BF16, 4, 256, out.data_ptr(), 256, None) < 0      # ld < D
    assert lib.asd_rmsnorm(None, 256, w.data_ptr(), 1e-6, Bd.DTYPE_BF16, 4, 256, out.data_ptr(), 256, None) < 0
    assert lib.asd_rmsnorm(x.data_ptr(), 256, w.data_ptr(), 1e-6, Bd.DTYPE_BF16, 4, 16384, out.data_ptr(), 16384, None) < 0   # D > 8192
    assert lib.asd_rmsnorm(x.data_ptr(), 256, w.data_ptr(), 1e-6, Bd.DTYPE_BF16, 0, 256, out.data_ptr(), 256, None) == 0      # nothing to do
    kc = torch.zeros(1, 1, 32, 128, dtype=BF, device="cuda")
    vt = torch.zeros(1, 1, 128, 32, dtype=BF, device="cuda")
    qkv = torch.zeros(1, 3 * 128, dtype=BF, device="cuda")
    pos = torch.zeros(1, dtype=torch.int32, device="cuda")
    args = (qkv.data_ptr(), 384, kc.data_ptr(), vt.data_ptr(), pos.data_ptr(), None, Bd.DTYPE_BF16, 1, 1, 1, 1)
    o = torch.zeros(1, 128, dtype=BF, device="cuda")
    assert lib.asd_attn_ragged(*args, 128, 32, o.data_ptr(), 128, None) == 0
    assert lib.asd_attn_ragged(*args, 64, 32, o.data_ptr(), 128, None) < 0          # head_dim 64
    assert lib.asd_attn_ragged(*args, 128, 40, o.data_ptr(), 128, None) < 0         # t_max % 32
    assert lib.asd_attn_ragged(*args, 128, 32, o.data_ptr(), 64, None) < 0          # ld_out < H * 128
    import ctypes as C
    shp = Bd.DecoderShape(256, 2, 1, 128, 512, 1e-6, None, 32)
    assert lib.asd_decoder_scratch_bytes(C.byref(shp), 4) == 0                       # inv_freq missing
    inv = torch.ones(64, device="cuda")
    shp = Bd.DecoderShape(256, 2, 1, 128, 512, 1e-6, inv.data_ptr(), 32)
    need = lib.asd_decoder_scratch_bytes(C.byref(shp), 4)
    assert need > 0
    sc = torch.empty(need + 256, dtype=torch.uint8, device="cuda")
    base = (sc.data_ptr() + 255) // 256 * 256
    x4 = torch.zeros(4, 256, dtype=BF, device="cuda")
    p4 = torch.zeros(4, dtype=torch.int32, device="cuda")
    assert lib.asd_decoder_forward(None, 0, C.byref(shp), x4.data_ptr(), 256, p4.data_ptr(), None, 4, 1, None, None, 0, base, need, None) == 0
    assert lib.asd_decoder_forward(None, 1, C.byref(shp), x4.data_ptr(), 256, p4.data_ptr(), None, 4, 1, None, None, 0, base, need, None) < 0   # layers missing
    assert lib.asd_decoder_forward(None, 0, C.byref(shp), x4.data_ptr(), 256, p4.data_ptr(), None, 4, 1, w.data_ptr(), None, 0, base, need, None) < 0  # norm without out
    assert lib.asd_decoder_forward(None, 0, C.byref(shp), x4.data_ptr(), 256, p4.data_ptr(), None, 4, 1, None, None, 0, base, 16, None) < 0        # scratch too small


def test_hip_decoder_with_weights_relaid_in_place_gives_the_same_bits():
    SL = import_module("adaptive-speculative-decoding_amd.serving.synthetic_lm")
    shape = SL.LMShape("small256", 512, 2, 4, 2, 1024, vocab=1000, rope_theta=1.0e6)      # 1024 | 512 | 2048 | 512 rows: all % 256
    a = SL.SyntheticLM(shape, dtype=BF, device="cuda", seed=9)
    b = SL.SyntheticLM(shape, dtype=BF, device="cuda", seed=9)
    a.enable_hip_layers()
    b.enable_hip_layers(pack_weights=True)
    assert b._hip.packed and not a._hip.packed
    oa = _run_sequence(a, 3, 20, 3, 5, seed=2)
    ob = _run_sequence(b, 3, 20, 3, 5, seed=2)
    for x, y in zip(oa, ob):
        assert torch.equal(x, y)
    with pytest.raises(RuntimeError):
        b.forward(torch.zeros(1, 2, dtype=torch.int64, device="cuda"))            # the torch modules would read re-laid bytes
    with pytest.raises(RuntimeError):
        b.enable_hip_layers(False)
    odd = SL.SyntheticLM(_small_shape(), dtype=BF, device="cuda", seed=9)
    odd_shape_ok = all(n % 256 == 0 for n in (512, 512 + 2 * 2 * 128, 2 * 1024))
    assert odd_shape_ok                                                                  # (_small_shape packs too)
