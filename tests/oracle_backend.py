"""A backend built on the CPU oracle, for exercising the HOST control flow of the package (pipeline
loop, decoder, table, sharding logic) on a box without a GPU.

TEST INFRASTRUCTURE: lives under tests/, is installed with asd_amd.set_backend() by fixtures only,
and is never importable from the package.  Numbers it produces say nothing about the product."""
import numpy as np
import torch

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
        assert inv_temperature == 1.0, "given logits are sharded at temperature 1 in the gloo tests"
        import torch
        store, dt = self._store(logits_shard)
        B, K, V = store.shape
        msg = O.lse_partial(store.reshape(B * K, V), dt, tok.numpy(), B, K, V, v_offset)
        msg[..., 0] /= np.log(2.0)
        return torch.from_numpy(msg)

    def lm_head_partial(self, hidden, weight_shard, tok, v_offset, inv_temperature=1.0):
        """f64 product of the operands, scaled by inv_temperature, then the shard message (m2, s, g) in f64
        (g is the scaled logit: this ops' accept_from_partials does not rescale)."""
        import torch
        B, K = tok.shape
        x = (hidden.double().reshape(B * K, -1) @ weight_shard.double().T).numpy() * float(np.float32(inv_temperature))
        t = tok.numpy().reshape(-1).astype(np.int64) - v_offset
        m = x.max(axis=1)
        s = np.exp(x - m[:, None]).sum(axis=1)
        inside = (t >= 0) & (t < x.shape[1])
        g = np.where(inside, x[np.arange(B * K), np.clip(t, 0, x.shape[1] - 1)], -np.inf)
        msg = np.stack([m / np.log(2.0), s, g], axis=-1).reshape(B, K, 3)
        return torch.from_numpy(msg)

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
            acc = ((lp > -np.inf) & (lu <= lp - lp_d.numpy().astype(np.float64))).astype(np.uint8)
        n_acc = np.array([int(np.argmin(np.append(a, 0))) for a in acc], dtype=np.int32)
        bits = np.array([sum(int(a[k]) << k for k in range(K)) for a in acc], dtype=np.int64)
        return torch.from_numpy(lp.astype(np.float32)), torch.from_numpy(acc), torch.from_numpy(n_acc), torch.from_numpy(bits)


    # ---- the rest of a tier step (asd_amd.serving.hierarchy)
    def lm_head_verify(self, hidden, weight, tok, lp_d, u, inv_temperature=1.0):
        """f64 product of the operands as stored (bf16 or f32), then the A5 rule on x * inv_temperature."""
        B, K = tok.shape
        x = hidden.double().reshape(B * K, -1).numpy() @ weight.double().numpy().T
        a = float(np.float32(inv_temperature))
        lp, acc, n_acc = O.py_verify_accept((x * a).reshape(B, K, -1), tok.numpy(), lp_d.numpy(), u.numpy())
        bits = np.array([sum(int(f) << k for k, f in enumerate(row)) for row in acc], dtype=np.int64)
        return (torch.from_numpy(lp.astype(np.float32)), torch.from_numpy(acc), torch.from_numpy(n_acc), torch.from_numpy(bits))

    def pack_predictor(self, predictor, device):
        return predictor.weights_numpy()

    def predictor_stop(self, pred, lp, feat, p_hist, stage_idx, costs, lam, risk_adjustment=True, n_obs=100, alpha=1.0,
                       beta=1.0, stats_col=5):
        score, k_star, hist = oracle_predictor_stop(pred, lp.numpy(), feat.numpy(), p_hist.numpy(), stage_idx,
                                                    costs.numpy(), lam, risk_adjustment, n_obs, alpha, beta, stats_col)
        p_hist.copy_(torch.from_numpy(hist))
        return torch.from_numpy(score), torch.from_numpy(k_star), p_hist

    def verify_stop(self, logits, tok, lp_d, u, inv_temperature, pred, feat, p_hist, stage_idx, costs, lam,
                    risk_adjustment=True, n_obs=100, alpha=1.0, beta=1.0, stats_col=5):
        v = self.verify_accept(logits, tok, lp_d, u, inv_temperature)
        return v, self.predictor_stop(pred, v[0], feat, p_hist, stage_idx, costs, lam, risk_adjustment, n_obs, alpha, beta,
                                      stats_col)

    def draft_sample(self, logits, r, inv_temperature=1.0, top_p=1.0):
        store, dt = _np_store(logits)
        B, V = store.shape
        d = O.draft_sample(store, dt, r.numpy(), B, V, inv_temperature, top_p)
        return torch.from_numpy(d["tok"]), torch.from_numpy(d["lp"].astype(np.float32)), torch.from_numpy(d["thr"])

    def residual_sample(self, t_logits, d_logits, n_acc, r, bonus, inv_temperature=1.0, d_threshold=None):
        st, dt = _np_store(t_logits)
        sd, _ = _np_store(d_logits)
        sb = None if bonus is None else _np_store(bonus)[0]
        B, K, V = st.shape
        tok, _ = O.residual_sample(st.reshape(B * K, V), sd.reshape(B * K, V), dt, n_acc.numpy(), r.numpy(), B, K, V,
                                   bonus=sb, inv_temperature=inv_temperature,
                                   d_threshold=None if d_threshold is None else d_threshold.numpy())
        return torch.from_numpy(tok)

    def commit_step(self, tok, n_acc, drawn, seq_len, tokens, n_commit, max_len):
        lens, out, nc = O.commit_step(tok.numpy(), n_acc.numpy(), drawn.numpy(), seq_len.numpy(), tokens.numpy(), max_len)
        seq_len.copy_(torch.from_numpy(lens))
        tokens.copy_(torch.from_numpy(out))
        n_commit.copy_(torch.from_numpy(nc))

    def lambda_sweep(self, p_hist, costs, lams):
        return torch.from_numpy(O.lambda_sweep(p_hist.numpy(), costs.numpy(), lams.numpy())[0])


# ---- the rest of a tier step (asd_amd.serving.hierarchy), by the oracle on CPU tensors ----------------------
def _np_store(t):
    import torch
    if t.dtype in (torch.bfloat16, torch.float16):
        return t.contiguous().view(torch.int16).numpy().view(np.uint16), (O.DT_BF16 if t.dtype == torch.bfloat16 else O.DT_F16)
    return t.float().contiguous().numpy(), O.DT_F32


def oracle_predictor_stop(weights, lp, feat, p_hist, stage_idx, costs, lam, risk_adjustment=True, n_obs=100, alpha=1.0,
                          beta=1.0, stats_col=5):
    """asd_predictor_stop restated with the oracle's pieces (numpy in / numpy out): A7 stats -> feature overlay ->
    A8 MLP -> A2 Bayes -> p_hist[:, stage_idx] -> A1 DP rule over all L stages."""
    w1, b1, w2, b2 = weights
    lp = np.asarray(lp, dtype=np.float32)
    x = np.array(feat, dtype=np.float32, copy=True)
    stats = O.logprob_stats(lp, None, K=lp.shape[1])
    if stats_col >= 0:
        x[:, stats_col:stats_col + 5] = stats.astype(np.float32)
    score = O.mlp_predict(x, w1, b1, np.asarray(w2).reshape(-1), b2)
    p = score.astype(np.float64)
    if risk_adjustment:
        p = O.bayes_adjust(p, n_obs, alpha, beta)
    hist = np.array(p_hist, dtype=np.float64, copy=True)
    hist[:, stage_idx] = p
    k_star, _ = O.optimal_stopping(hist, np.asarray(costs, dtype=np.float64), float(lam))
    return score, k_star, hist
