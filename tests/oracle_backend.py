"""A backend built on the CPU oracle, for exercising the HOST control flow of the package (pipeline
loop, decoder, table, sharding logic) on a box without a GPU.

TEST INFRASTRUCTURE: lives under tests/, is installed with asd_amd.set_backend() by fixtures only,
and is never importable from the package.  Numbers it produces say nothing about the product."""
import numpy as np

from oracle import oracle as O


class OracleBackend:
    name = "oracle-cpu (tests only)"

    def optimal_stopping(self, p, Cc, lam, risk_adjustment=False, alpha=1.0, beta=1.0):
        return O.optimal_stopping(p, Cc, lam, risk_adjustment, alpha, beta)

    def bayes_adjust(self, p, n_obs, alpha=1.0, beta=1.0):
        return O.bayes_adjust(p, n_obs, alpha, beta)

    def expected_cost(self, p, Cc, lam, k):
        return O.expected_cost(p, Cc, lam, k)

    def derive_thresholds(self, q, c, lam):
        return O.derive_thresholds(q, c, lam)[0]

    def lambda_sweep(self, p, Cc, lam, risk_adjustment=False, alpha=1.0, beta=1.0):
        return O.lambda_sweep(p, Cc, lam, risk_adjustment, alpha, beta)

    def mlp_predict(self, x, w1, b1, w2, b2):
        return O.mlp_predict(x, w1, b1, np.asarray(w2).reshape(-1), b2)

    def threshold_stop(self, score, theta):
        return O.threshold_stop(score, theta)

    def logprob_stats(self, lp, n_valid=None):
        lp = np.asarray(lp, dtype=np.float32)
        return O.logprob_stats(lp, n_valid, K=lp.shape[-1])

    def verify_accept(self, logits, tok, lp_draft, u):
        import torch
        if isinstance(logits, torch.Tensor):
            if logits.dtype == torch.bfloat16:
                store, dt = logits.contiguous().view(torch.int16).numpy().view(np.uint16), O.DT_BF16
            elif logits.dtype == torch.float16:
                store, dt = logits.contiguous().view(torch.int16).numpy().view(np.uint16), O.DT_F16
            else:
                store, dt = logits.float().contiguous().numpy(), O.DT_F32
        else:
            store, dt = np.ascontiguousarray(logits, dtype=np.float32), O.DT_F32
        B, K, V = store.shape
        r = O.verify_accept(store.reshape(B * K, V), dt, tok, lp_draft, u, B, K, V)
        return dict(lp_t=r["lp_t"], accept=r["accept"], n_acc=r["n_acc"], bits=r["bits"])


class OracleOps:
    """`ops` object for asd_amd.distributed on CPU tensors (gloo tests): same protocol as HipOps,
    arithmetic by the oracle.  msg = (m2, s, g) float64 with sum_v exp(x) = s * 2^m2."""

    @staticmethod
    def _store(logits):
        import torch
        if logits.dtype == torch.bfloat16:
            return logits.contiguous().view(torch.int16).numpy().view(np.uint16), O.DT_BF16
        if logits.dtype == torch.float16:
            return logits.contiguous().view(torch.int16).numpy().view(np.uint16), O.DT_F16
        return logits.float().contiguous().numpy(), O.DT_F32

    def verify_accept(self, logits, tok, lp_d, u, inv_temperature=1.0):
        import torch
        store, dt = self._store(logits)
        B, K, V = store.shape
        r = O.verify_accept(store.reshape(B * K, V), dt, tok.numpy(), lp_d.numpy(), u.numpy(), B, K, V,
                            inv_temperature=inv_temperature)
        return (torch.from_numpy(r["lp_t"]), torch.from_numpy(r["accept"]), torch.from_numpy(r["n_acc"]),
                torch.from_numpy(r["bits"].view(np.int64)))

    def lse_partial(self, logits_shard, tok, v_offset, inv_temperature=1.0):
        assert inv_temperature == 1.0, "the gloo tests shard at temperature 1"
        import torch
        store, dt = self._store(logits_shard)
        B, K, V = store.shape
        msg = O.lse_partial(store.reshape(B * K, V), dt, tok.numpy(), B, K, V, v_offset)
        msg[..., 0] /= np.log(2.0)
        return torch.from_numpy(msg)

    def lm_head_partial(self, hidden, weight_shard, tok, v_offset, inv_temperature=1.0):
        """f64 product of the bf16 operands, then the shard message of lse_partial on those logits."""
        import torch
        x = hidden.double().reshape(-1, hidden.shape[-1]) @ weight_shard.double().T
        B, K = tok.shape
        return self.lse_partial(x.float().reshape(B, K, -1), tok, v_offset, inv_temperature)

    def accept_from_partials(self, msg_all, lp_d, u, inv_temperature=1.0):
        import torch
        m = msg_all.numpy().astype(np.float64)
        R, B, K, _ = m.shape
        with np.errstate(all="ignore"):
            M = m[..., 0].max(axis=0)
            s = (m[..., 1] * np.exp2(m[..., 0] - M[None])).sum(axis=0)
            lse = np.log(2.0) * (M + np.log2(s))
            g = m[..., 2].max(axis=0)
            lp = g - lse
            uu = u.numpy().astype(np.float64)
            lu = np.where(uu > 0, np.log(np.where(uu > 0, uu, 1.0)), -np.inf)
            acc = (lu <= lp - lp_d.numpy().astype(np.float64)).astype(np.uint8)
        n_acc = np.array([int(np.argmin(np.append(a, 0))) for a in acc], dtype=np.int32)
        bits = np.array([sum(int(a[k]) << k for k in range(K)) for a in acc], dtype=np.int64)
        return torch.from_numpy(lp.astype(np.float32)), torch.from_numpy(acc), torch.from_numpy(n_acc), torch.from_numpy(bits)
