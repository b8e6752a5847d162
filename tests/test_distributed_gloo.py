"""The N > 1 exchange logic of asd_amd.distributed on CPU: world_size 2, gloo, 127.0.0.1, with the
oracle as `ops` (tests/oracle_backend.py: OracleOps).  Checks that the sharded forms give the
single-process oracle answer and that only the small messages cross ranks.  The multi-rank token-level LOOP
(draft / 32B / 72B roles, stop-or-escalate) is covered by tests/test_hierarchy.py."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _case(B=6, K=5, V=1003, seed=3):
    sys.path.insert(0, ROOT)
    from tests.helpers import make_verify_case
    from oracle import oracle as O
    return make_verify_case(B, K, V, O.DT_BF16, seed=seed, n_threads=1)


def _worker(rank, world, port, what, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from asd_amd import distributed as D
        from tests.oracle_backend import OracleOps
        case = _case()
        B, K, V = case["B"], case["K"], case["V"]
        lg = torch.from_numpy(case["logits"].view(np.int16)).view(torch.bfloat16).view(B, K, V)
        tok, lp_d, u = (torch.from_numpy(case[k]) for k in ("tok", "lp_d", "u"))
        ref = case["ref"]
        if what == "vocab":
            v = D.VocabShardedVerifier(V, ops=OracleOps())
            assert (v.v0, v.v1) == D.shard_bounds(V, world, rank)
            lp, acc, n_acc, bits = v.verify(lg[:, :, v.v0:v.v1].contiguous(), tok, lp_d, u)
            np.testing.assert_allclose(lp.numpy(), ref["lp_t64"], atol=1e-5, rtol=1e-6)
            assert np.array_equal(acc.numpy(), ref["accept"]) and np.array_equal(n_acc.numpy(), ref["n_acc"])
            assert np.array_equal(bits.numpy().view(np.uint64), ref["bits"])
            with pytest.raises(ValueError):
                v.verify(lg, tok, lp_d, u)
        elif what == "vocab_hidden":
            # tensor-parallel lm_head: hidden states replicated, weight rows split; only [B,K,3] crosses the link
            g = torch.Generator().manual_seed(5)
            Dm = 64
            hid = torch.randn((B, K, Dm), generator=g).to(torch.bfloat16)
            wgt = (torch.randn((V, Dm), generator=g) * (3.0 / Dm ** 0.5)).to(torch.bfloat16)
            hb = hid.reshape(B * K, Dm).view(torch.int16).numpy().view(np.uint16)
            wb = wgt.view(torch.int16).numpy().view(np.uint16)
            from oracle import oracle as O
            full = O.lm_head_verify(hb, wb, tok.numpy(), lp_d.numpy(), u.numpy(), B, K)
            v = D.VocabShardedVerifier(V, ops=OracleOps())
            lp, acc, n_acc, bits = v.verify_hidden(hid, wgt[v.v0:v.v1].contiguous(), tok, lp_d, u)
            np.testing.assert_allclose(lp.numpy(), full["lp_t64"], atol=2e-5, rtol=1e-6)
            safe = full["margin"] >= 1e-4
            assert np.array_equal(acc.numpy()[safe], full["accept"][safe])
            with pytest.raises(ValueError):
                v.verify_hidden(hid, wgt, tok, lp_d, u)
        elif what == "batch":
            v = D.BatchShardedVerifier(B, ops=OracleOps())
            s = slice(v.b0, v.b1)
            lp, acc, n_acc, bits = v.verify_local(lg[s].contiguous(), tok[s].contiguous(), lp_d[s].contiguous(),
                                                  u[s].contiguous())
            assert np.array_equal(acc.numpy(), ref["accept"][s])
            full = v.gather_n_acc(n_acc)
            assert np.array_equal(full.numpy(), ref["n_acc"])
        elif what == "tiers":
            link = D.TierLink(draft_rank=0, target_rank=1)
            if rank == 0:                       # draft tier: owns tok / lp_d, never sees logits
                link.send_draft(tok, lp_d)
                acc, n_acc = link.recv_verdict(B, K, "cpu")
                assert np.array_equal(acc.numpy(), ref["accept"]) and np.array_equal(n_acc.numpy(), ref["n_acc"])
            else:                               # target tier: owns logits and u
                t2, l2 = link.recv_draft(B, K, "cpu")
                assert torch.equal(t2, tok) and torch.equal(l2, lp_d)
                lp, acc, n_acc, bits = OracleOps().verify_accept(lg, t2, l2, u)
                link.send_verdict(acc, n_acc)
        ret[rank] = "ok"
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("what", ["vocab", "vocab_hidden", "batch", "tiers"])
def test_two_rank_gloo(what):
    ctx = mp.get_context("spawn")
    with ctx.Manager() as mgr:
        ret = mgr.dict()
        procs = [ctx.Process(target=_worker, args=(r, 2, port, what, ret)) for port in [_free_port()] for r in range(2)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(120)
        assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
        assert dict(ret) == {0: "ok", 1: "ok"}


def test_shard_bounds_cover_everything():
    sys.path.insert(0, ROOT)
    from asd_amd.distributed import shard_bounds
    for total in (0, 1, 7, 152064):
        for world in (1, 2, 3, 8):
            edges = [shard_bounds(total, world, r) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(edges, edges[1:]))
            assert max(e - s for s, e in edges) - min(e - s for s, e in edges) <= 1
