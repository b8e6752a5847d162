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
