"""Re-entrancy of the C ABI and of the Python front end ("the launcher must be
re-entrant").  The reference calls its pipeline from a pool of up to 100 threads (src/serving/pipeline.py:83,144-163):
here 16 host threads hammer the library at once -- raw asd_verify_accept through ctypes, HipOps.verify_stop (ONE shared
HipOps: its scratch is per thread), the lm_head path with its shared packed image, the draft sampler and
pipeline.process_request -- each thread on its own HIP stream with its own workspace.  Every result must equal the one the
same call gives when it runs alone."""
import threading

import numpy as np
import pytest

from oracle import oracle as O
from tests.helpers import make_verify_case, to_device_logits

pytestmark = pytest.mark.gpu

N_THREADS, ITERS = 16, 12
SHAPES = [(8, 8, 20000), (32, 8, 30000), (5, 4, 30000), (40, 8, 9000)]      # split rows, one workgroup per row, tiny, > CUs


def test_sixteen_threads_on_their_own_streams_get_the_serial_results(golden):
    import torch
    import asd_amd
    from asd_amd import kernels as K
    from asd_amd.distributed import HipOps
    from asd_amd.minimal_adaptive_decoder import MinimalQualityPredictor
    from asd_amd.serving import AdaptiveSpeculativePipeline, PipelineConfig
    from tests.test_host_logic import FakeStageManager, ScriptedPredictor
    asd_amd.set_backend(None)
    lib = K._lib()
    ops = HipOps()
    torch.manual_seed(0)
    pred = ops.pack_predictor(MinimalQualityPredictor().eval(), torch.device("cuda"))
    Cc = torch.tensor([1.0, 4.5, 10.0], dtype=torch.float64, device="cuda")
    pipe = AdaptiveSpeculativePipeline(FakeStageManager(), ScriptedPredictor(), object(),
                                       PipelineConfig(lambda_value=30.0, stop_rule="full", risk_adjustment=True))
    g = torch.Generator(device="cuda").manual_seed(3)
    w = (torch.randn((3000, 128), generator=g, device="cuda") * (3.0 / 128 ** 0.5)).to(torch.bfloat16)     # one lm_head, shared
    jobs = []
    for i in range(N_THREADS):
        B, Kk, V = SHAPES[i % len(SHAPES)]
        c = make_verify_case(B, Kk, V, O.DT_BF16, seed=1000 + i)
        lg = to_device_logits(c["logits"], c["dtype"]).view(B, Kk, V)
        tok, lp_d, u = (torch.from_numpy(c[k]).cuda() for k in ("tok", "lp_d", "u"))
        feat = torch.from_numpy((np.random.default_rng(i).standard_normal((B, 64)) * 0.3).astype(np.float32)).cuda()
        hid = torch.randn((B, Kk, 128), generator=g, device="cuda").to(torch.bfloat16)
        tok_h = torch.randint(0, 3000, (B, Kk), generator=g, device="cuda", dtype=torch.int32)
        rows = (torch.randn((B, V), generator=g, device="cuda") * 3).to(torch.bfloat16)
        r = torch.rand((B,), generator=g, device="cuda")
        jobs.append(dict(c=c, lg=lg, tok=tok, lp_d=lp_d, u=u, feat=feat, hid=hid, tok_h=tok_h, rows=rows, r=r,
                         prompt=("easy", "hard", "mid")[i % 3] + f" request {i}"))
    torch.cuda.synchronize()

    def run(j, raw_ws=None, out=None):
        B, Kk, V = j["c"]["B"], j["c"]["K"], j["c"]["V"]
        st = torch.cuda.current_stream().cuda_stream
        if raw_ws is None:
            raw_ws = K.VerifyWorkspace(B, Kk, V)
            out = K.VerifyResult(torch.empty((B, Kk), dtype=torch.float32, device="cuda"), torch.empty((B, Kk), dtype=torch.uint8, device="cuda"),
                                 torch.empty((B,), dtype=torch.int32, device="cuda"), torch.empty((B,), dtype=torch.int64, device="cuda"))
        rc = lib.asd_verify_accept(j["lg"].data_ptr(), 1, V, j["tok"].data_ptr(), j["lp_d"].data_ptr(), j["u"].data_ptr(), B, Kk, V,
                                   out.lp_target.data_ptr(), out.accept.data_ptr(), out.n_acc.data_ptr(), out.accept_bits.data_ptr(),
                                   raw_ws.buf.data_ptr(), raw_ws.bytes, st)
        assert rc == 0
        ph = torch.ones((B, 3), dtype=torch.float64, device="cuda")
        v, s = ops.verify_stop(j["lg"], j["tok"], j["lp_d"], j["u"], 1.0, pred, j["feat"], ph, 1, Cc, 0.8)
        lmh = ops.lm_head_verify(j["hid"], w, j["tok_h"], j["lp_d"], j["u"])
        d = ops.draft_sample(j["rows"], j["r"], float(np.float32(1 / 0.7)), 0.9)
        res = pipe.process_request(j["prompt"])
        torch.cuda.current_stream().synchronize()
        return dict(raw=(out.lp_target.clone(), out.accept.clone(), out.n_acc.clone()), v=[t.clone() for t in v],
                    s=(s[0].clone(), s[1].clone(), ph), lmh=[t.clone() for t in lmh], d=[t.clone() for t in d],
                    # (the Bayes prior count is max(100, requests served so far), pipeline.py:236: the probabilities drift in the
                    # 4th digit once the pool has served 100 requests; the stop decision of these prompts does not)
                    req=(res.stopped_at_stage, len(res.stage_probabilities)), ws=raw_ws, out=out)

    serial = [run(j) for j in jobs]
    for j, sres in zip(jobs, serial):
        assert np.array_equal(sres["raw"][1].cpu().numpy(), j["c"]["ref"]["accept"])
    errors = []
    gate = threading.Barrier(N_THREADS)

    def same(a, b):
        if isinstance(a, torch.Tensor):
            return torch.equal(a, b)
        if isinstance(a, (list, tuple)):
            return len(a) == len(b) and all(same(x, y) for x, y in zip(a, b))
        return a == b

    def body(i):
        try:
            stream = torch.cuda.Stream()
            gate.wait(timeout=120)
            with torch.cuda.stream(stream):
                for it in range(ITERS):
                    got = run(jobs[i], serial[i]["ws"], serial[i]["out"])
                    for key in ("raw", "v", "s", "lmh", "d", "req"):
                        if not same(got[key], serial[i][key]):
                            raise AssertionError(f"thread {i} iteration {it}: `{key}` differs from the serial result")
        except Exception as e:  # noqa: BLE001
            errors.append(f"{type(e).__name__}: {e}")

    threads = [threading.Thread(target=body, args=(i,)) for i in range(N_THREADS)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(600)
    pipe.shutdown()
    assert not any(t.is_alive() for t in threads) and not errors, errors[:3]
    assert len(ops._packed) == 1, "one packed image of the shared lm_head, whatever the number of threads"


def test_packed_lm_head_follows_an_in_place_weight_update():
    """HipOps keys the packed image on (address, shape, stride, dtype, torch's version counter): after `w.add_` the packed
    call must see the NEW weights (the address alone served the stale image), and a row slice that shares the
    address is its own matrix."""
    import torch
    from asd_amd import kernels as K
    from asd_amd.distributed import HipOps
    g = torch.Generator(device="cuda").manual_seed(5)
    V, D, B, Kk = 4096, 256, 4, 8
    w = (torch.randn((V, D), generator=g, device="cuda") * (3.0 / D ** 0.5)).to(torch.bfloat16)
    hid = torch.randn((B, Kk, D), generator=g, device="cuda").to(torch.bfloat16)
    tok = torch.randint(0, V // 2, (B, Kk), generator=g, device="cuda", dtype=torch.int32)
    lp_d = -torch.rand((B, Kk), generator=g, device="cuda")
    u = torch.rand((B, Kk), generator=g, device="cuda")
    packed_ops, plain_ops = HipOps(pack_lm_head=True), HipOps(pack_lm_head=False)

    def both(weight):
        a = packed_ops.lm_head_verify(hid, weight, tok, lp_d, u)
        b = plain_ops.lm_head_verify(hid, weight, tok, lp_d, u)
        torch.cuda.synchronize()
        return a, b
    a, b = both(w)
    assert all(torch.equal(x, y) for x, y in zip(a, b))
    before = a[0].clone()
    with torch.no_grad():
        w.add_((torch.randn(w.shape, generator=g, device="cuda") * 0.05).to(w.dtype))      # in place: same data_ptr
    a, b = both(w)
    assert all(torch.equal(x, y) for x, y in zip(a, b)), "the packed call used a stale image"
    assert not torch.equal(a[0], before)
    half = w[: V // 2]                                                                   # same address, another matrix
    a, b = both(half)
    assert all(torch.equal(x, y) for x, y in zip(a, b))
    assert len(packed_ops._packed) == 2 and {k[1] for k in packed_ops._packed} == {(V, D), (V // 2, D)}


def test_get_stats_reports_the_last_verify_step_and_roctx_ranges_are_live():
    """get_stats() keeps the reference's keys (pipeline.py:346-370) and gains `kernel_us` / `hbm_gbps` of the last
    verify step of an attached, profiling HipOps; the hot-path calls run inside roctx ranges."""
    import torch
    from asd_amd import trace
    from asd_amd.distributed import HipOps
    from asd_amd.serving import AdaptiveSpeculativePipeline, PipelineConfig
    from tests.test_host_logic import FakeStageManager, ScriptedPredictor
    assert trace.enabled()                                       # torch.cuda.nvtx == roctx on this build
    c = make_verify_case(32, 8, 152064, O.DT_BF16, seed=4)
    lg = to_device_logits(c["logits"], c["dtype"]).view(32, 8, 152064)
    tok, lp_d, u = (torch.from_numpy(c[k]).cuda() for k in ("tok", "lp_d", "u"))
    ops = HipOps(profile=True)
    pipe = AdaptiveSpeculativePipeline(FakeStageManager(), ScriptedPredictor(), object(), PipelineConfig())
    assert "kernel_us" not in pipe.get_stats()
    pipe.attach_hot_path(ops)
    for _ in range(50):
        out = ops.verify_accept(lg, tok, lp_d, u)
    st = pipe.get_stats()
    for key in ("total_requests", "stage_distribution", "avg_tokens_per_request", "active_requests"):     # the reference's keys
        assert key in st
    assert 5.0 < st["kernel_us"] < 200.0 and 400.0 < st["hbm_gbps"] < 8000.0, st
    assert np.array_equal(out[1].cpu().numpy(), c["ref"]["accept"])
    pipe.shutdown()


def test_stats_report_the_calling_threads_own_step():
    """HipOps documents itself re-entrant (one scratch set per calling thread): the profiled event pair is per thread too, so
    stats() never reports another thread's step (round 3 kept ONE slot for all threads)."""
    import threading
    import torch
    from asd_amd.distributed import HipOps
    ops = HipOps(profile=True)
    shapes = {"a": (8, 8, 32000), "b": (32, 4, 152064)}
    got, errs = {}, []
    barrier = threading.Barrier(2)

    def work(name):
        try:
            Bn, Kn, Vn = shapes[name]
            c = make_verify_case(Bn, Kn, Vn, O.DT_BF16, seed=11)
            lg = to_device_logits(c["logits"], c["dtype"]).view(Bn, Kn, Vn)
            tok, lp_d, u = (torch.from_numpy(c[k]).cuda() for k in ("tok", "lp_d", "u"))
            with torch.cuda.stream(torch.cuda.Stream()):
                for _ in range(20):
                    ops.verify_accept(lg, tok, lp_d, u)
                barrier.wait(timeout=60)                 # both threads have profiled their last step before either reads
                got[name] = ops.stats()
        except Exception as e:  # noqa: BLE001
            errs.append((name, repr(e)))
    ts = [threading.Thread(target=work, args=(n,)) for n in shapes]
    for t in ts:
        t.start()
    for t in ts:
        t.join(120)
    assert not errs, errs
    for name, (Bn, Kn, Vn) in shapes.items():
        assert got[name]["algorithmic_bytes"] == Bn * Kn * Vn * 2 + 17 * Bn * Kn + 12 * Bn, (name, got[name])
        assert got[name]["kernel_us"] > 0
    assert ops.stats() == {}                             # this (main) thread profiled nothing
