"""CPU oracle for the hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module, and
only as the checker / timed CPU baseline.  Nothing under adaptive-speculative-decoding_amd/ imports
it; the product path is libasd_hip.so and fails loudly without it.

Two layers:
  * `lib`  -- ctypes view of oracle/_build/libasd_oracle.so (asd_oracle.c, plain C, f64 where the
              reference is CPython float arithmetic);
  * `py_*` -- pure-Python/numpy restatements that follow the reference line by line; they are
              slow and are used on small cases to cross-check the C restatement.

Pinning (tests/test_oracle_golden.py): every function with a reference symbol is checked against
tests/golden/*.npz, produced by oracle/gen_golden.py from the reference's own files.  The
token-level accept test (A5) has no reference symbol: PARITY UNPINNED for it.
All citations are relative to /root/reference.
"""
from __future__ import annotations

import ctypes as C
import math
import os
import subprocess
from typing import List, Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libasd_oracle.so")

DT_F32, DT_BF16, DT_F16 = 0, 1, 2


def build(force: bool = False) -> str:
    """Compile asd_oracle.c with gcc (oracle/Makefile)."""
    if force or not os.path.exists(_LIB_PATH) or (
        os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "asd_oracle.c"))
    ):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _LIB_PATH


_lib = None


def _load():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
    return _lib


def _p(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _c(a, dtype) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=dtype)


# ----------------------------------------------------------------------------- bf16 / f16 helpers
def f32_to_bf16_bits(x: np.ndarray) -> np.ndarray:
    """Round-to-nearest-even f32 -> bf16 bit pattern (uint16); NaN stays NaN."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    u = x.view(np.uint32).astype(np.uint64)
    rounded = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)
    nan = np.isnan(x)
    if nan.any():
        rounded = np.where(nan, np.uint16(0x7FC0), rounded)
    return rounded


def bf16_bits_to_f32(b: np.ndarray) -> np.ndarray:
    return (np.ascontiguousarray(b, dtype=np.uint16).astype(np.uint32) << 16).view(np.float32)


def logits_as_f32(logits: np.ndarray, dtype: int) -> np.ndarray:
    if dtype == DT_F32:
        return np.asarray(logits, dtype=np.float32)
    if dtype == DT_BF16:
        return bf16_bits_to_f32(logits)
    return np.asarray(logits).view(np.float16).astype(np.float32)


# ----------------------------------------------------------------------------- A5 / A6
def verify_accept(logits: np.ndarray, dtype: int, tok, lp_d, u, B: int, K: int, V: int,
                  ld_row: Optional[int] = None, n_threads: int = 1, inv_temperature: float = 1.0):
    """A5 (+A6).  logits: uint16 bit patterns for bf16/f16, float32 for f32; shape [B*K, ld_row].

    Returns dict(lp_t f32[B,K], lp_t64 f64[B,K], accept u8[B,K], n_acc i32[B], bits u64[B],
    margin f64[B,K])."""
    lib = _load()
    logits = np.ascontiguousarray(logits)
    if ld_row is None:
        ld_row = V
    assert logits.size >= (B * K - 1) * ld_row + V if B * K > 0 else True
    tok = _c(tok, np.int32).reshape(-1)
    lp_d = _c(lp_d, np.float32).reshape(-1)
    u = _c(u, np.float32).reshape(-1)
    lp_t = np.empty(B * K, np.float32)
    lp_t64 = np.empty(B * K, np.float64)
    margin = np.empty(B * K, np.float64)
    acc = np.empty(B * K, np.uint8)
    n_acc = np.empty(B, np.int32)
    bits = np.empty(B, np.uint64)
    rc = lib.oracle_verify_accept(_p(logits), C.c_int(dtype), C.c_int64(ld_row), _p(tok), _p(lp_d),
                                  _p(u), C.c_int(B), C.c_int(K), C.c_int(V), _p(lp_t), _p(acc),
                                  _p(n_acc), _p(bits), _p(lp_t64), _p(margin), C.c_int(n_threads),
                                  C.c_float(inv_temperature))
    if rc != 0:
        raise ValueError(f"oracle_verify_accept rc={rc}")
    return dict(lp_t=lp_t.reshape(B, K), lp_t64=lp_t64.reshape(B, K), accept=acc.reshape(B, K),
                n_acc=n_acc, bits=bits, margin=margin.reshape(B, K))


def lse_partial(logits: np.ndarray, dtype: int, tok, B: int, K: int, V_shard: int, v_offset: int,
                ld_row: Optional[int] = None) -> np.ndarray:
    lib = _load()
    logits = np.ascontiguousarray(logits)
    if ld_row is None:
        ld_row = V_shard
    tok = _c(tok, np.int32).reshape(-1)
    msg = np.empty((B, K, 3), np.float64)
    rc = lib.oracle_lse_partial(_p(logits), C.c_int(dtype), C.c_int64(ld_row), _p(tok), C.c_int(B),
                                C.c_int(K), C.c_int(V_shard), C.c_int64(v_offset), _p(msg))
    if rc != 0:
        raise ValueError(rc)
    return msg


def residual_sample(t_logits: np.ndarray, d_logits: np.ndarray, dtype: int, n_acc, r, B: int, K: int, V: int,
                    bonus: Optional[np.ndarray] = None, inv_temperature: float = 1.0, d_threshold=None):
    """asd_residual_sample in f64.  t_logits / d_logits: storage arrays [B*K, V]; bonus: [B, V] or None;
    d_threshold: [B, K] f32 nucleus thresholds of the draft rows (asd_draft_sample) or None.
    Returns (token i32[B], margin f64[B])."""
    lib = _load()
    t_logits, d_logits = np.ascontiguousarray(t_logits), np.ascontiguousarray(d_logits)
    bonus = None if bonus is None else np.ascontiguousarray(bonus)
    n_acc = _c(n_acc, np.int32).reshape(-1)
    r = _c(r, np.float32).reshape(-1)
    thr = None if d_threshold is None else _c(d_threshold, np.float32).reshape(-1)
    tok = np.empty(B, np.int32)
    margin = np.empty(B, np.float64)
    rc = lib.oracle_residual_sample(_p(t_logits), C.c_int64(V), _p(d_logits), C.c_int64(V), _p(bonus), C.c_int64(V),
                                    C.c_int(dtype), _p(n_acc), _p(r), C.c_int(B), C.c_int(K), C.c_int(V),
                                    C.c_float(inv_temperature), _p(tok), _p(margin), _p(thr))
    if rc != 0:
        raise ValueError(rc)
    return tok, margin


def draft_sample(logits: np.ndarray, dtype: int, r, B: int, V: int, inv_temperature: float = 1.0, top_p: float = 1.0,
                 ld_row: Optional[int] = None):
    """asd_draft_sample in f64 (X1; the reference delegates it to HF generate with temperature / top_p,
    generate_training_data.py:110-119: the truncated distribution is pinned to HF's TemperatureLogitsWarper +
    TopPLogitsWarper by tests/golden/top_p_nucleus.npz, the draw -- inverse CDF in vocabulary order -- is this build's own
    rule, PARITY UNPINNED).  logits: storage array [B, ld_row].
    Returns dict(tok i32[B], lp f64[B], thr f32[B] nucleus threshold logit (-inf = no truncation),
    margin_p f64[B] (distance of top_p to the bracketing cumulative masses), margin_r f64[B] (CDF-edge distance))."""
    lib = _load()
    logits = np.ascontiguousarray(logits)
    if ld_row is None:
        ld_row = V
    r = _c(r, np.float32).reshape(-1)
    tok = np.empty(B, np.int32)
    lp = np.empty(B, np.float64)
    thr = np.empty(B, np.float32)
    mp = np.empty(B, np.float64)
    mr = np.empty(B, np.float64)
    rc = lib.oracle_draft_sample(_p(logits), C.c_int64(ld_row), C.c_int(dtype), _p(r), C.c_int(B), C.c_int(V),
                                 C.c_float(inv_temperature), C.c_float(top_p), _p(tok), _p(lp), _p(thr), _p(mp), _p(mr))
    if rc != 0:
        raise ValueError(rc)
    return dict(tok=tok, lp=lp, thr=thr, margin_p=mp, margin_r=mr)


def row_softmax_stats(logits: np.ndarray, dtype: int, R: int, V: int, inv_temperature: float = 1.0,
                      ld_row: Optional[int] = None) -> Tuple[np.ndarray, np.ndarray]:
    """asd_verify_accept_stats' two row statistics in f64: (max_v log softmax(x a)[v], -sum_v p_v ln p_v) per row, the
    quantities docs/guides/RESEARCH_PROTOCOL.md:378-400 calls np.max(lp) and -np.sum(np.exp(lp) * lp) with `lp` the
    complete log-prob vector of the position (doc only: not a parity target).  Rows without a finite logit: NaN."""
    ld = V if ld_row is None else ld_row
    a = float(np.float32(inv_temperature))
    flat = np.ascontiguousarray(logits).reshape(-1)
    mx = np.empty(R)
    ent = np.empty(R)
    for r in range(R):
        x = logits_as_f32(flat[r * ld:r * ld + V], dtype).astype(np.float64) * a
        m = x.max() if V else -np.inf
        if not np.isfinite(m):
            mx[r] = ent[r] = np.nan
            continue
        z = x - m
        e = np.exp(z)
        s = e.sum()
        lp = z - np.log(s)
        mx[r] = lp.max()
        pos = e > 0
        ent[r] = -(e[pos] / s * lp[pos]).sum()
    return mx, ent


def py_token_logprob_reference_idiom(score_row_f32: np.ndarray, token_id: int) -> float:
    """generate_training_data.py:131-133 restated with numpy in the row's own precision:
    probs = softmax(score[0]); logprob = log(probs[token_id])."""
    x = np.asarray(score_row_f32)
    e = np.exp(x - x.max())
    probs = e / e.sum()
    return float(np.log(probs[token_id]))


def py_verify_accept(logits_f32: np.ndarray, tok, lp_d, u):
    """numpy f64 restatement of A5 on [B,K,V] f32 logits (small cases)."""
    x = np.asarray(logits_f32, dtype=np.float64)
    B, K, V = x.shape
    tok = np.asarray(tok).reshape(B, K)
    with np.errstate(all="ignore"):
        m = x.max(axis=-1)
        s = np.exp(x - m[..., None]).sum(axis=-1)
        lse = np.where(np.isneginf(m), -np.inf, m + np.log(s))
        inside = (tok >= 0) & (tok < V)
        g = np.where(inside, np.take_along_axis(x, np.clip(tok, 0, V - 1)[..., None], -1)[..., 0],
                     -np.inf)
        lp = g - lse
        uu = np.asarray(u, dtype=np.float32).reshape(B, K).astype(np.float64)
        lu = np.where(uu > 0, np.log(np.where(uu > 0, uu, 1.0)), np.where(uu == 0, -np.inf, np.nan))
        acc = (lp > -np.inf) & (lu <= (lp - np.asarray(lp_d, dtype=np.float32).reshape(B, K).astype(np.float64)))
    n_acc = np.array([int(np.argmin(np.append(a, False))) for a in acc], dtype=np.int32)
    return lp, acc.astype(np.uint8), n_acc


def lm_head_verify(hidden_bits: np.ndarray, weight_bits: np.ndarray, tok, lp_d, u, B: int, K: int,
                   inv_temperature: float = 1.0):
    """N2 oracle: logits = hidden @ weight.T in f64 from the bf16 bit patterns (hidden [B*K, D],
    weight [V, D], both uint16), then the A5 rule on x * inv_temperature in f64.  The f32
    inv_temperature is used exactly as the C oracle does.  Parity unpinned (A5 has no reference
    symbol).  Returns dict(lp_t64, accept, n_acc, bits, margin, logits64)."""
    h = bf16_bits_to_f32(np.asarray(hidden_bits)).astype(np.float64)
    wb = np.asarray(weight_bits)
    V = wb.shape[0]
    x = np.empty((h.shape[0], V), np.float64)
    step = max(1, (1 << 27) // max(1, wb.shape[1]))     # <= 1 GiB of f64 weights at a time (a 152064 x 8192 head is 10 GB)
    for v0 in range(0, V, step):
        x[:, v0:v0 + step] = h @ bf16_bits_to_f32(wb[v0:v0 + step]).astype(np.float64).T
    a = float(np.float32(inv_temperature))
    lp, acc, n_acc = py_verify_accept((x * a).reshape(B, K, V), tok, lp_d, u)
    bits = np.array([sum(int(f) << k for k, f in enumerate(row)) for row in acc], dtype=np.uint64)
    with np.errstate(all="ignore"):
        uu = np.asarray(u, dtype=np.float32).reshape(B, K).astype(np.float64)
        margin = np.abs(np.log(uu) - (lp - np.asarray(lp_d, dtype=np.float32).reshape(B, K).astype(np.float64)))
    return dict(lp_t64=lp, accept=acc, n_acc=n_acc, bits=bits, margin=margin, logits64=x.reshape(B, K, V))


def lambda_sweep(p, Cc, lam, risk_adjustment: bool = False, alpha: float = 1.0, beta: float = 1.0):
    """N4 oracle: the DP rule (C restatement, pinned by dp_rule.npz) per lambda, then sum C[:k*+1] and
    prod p[:k*+1] left to right like dp_solver.py:92-98.  Returns (k_star [G,B] i32, cost [G,B], p_ok [G,B])."""
    p = np.ascontiguousarray(p, dtype=np.float64)
    Cc = np.ascontiguousarray(Cc, dtype=np.float64).reshape(-1)
    lam = np.ascontiguousarray(lam, dtype=np.float64).reshape(-1)
    B, L = p.shape
    ks = np.zeros((lam.size, B), np.int32)
    cost = np.zeros((lam.size, B))
    ok = np.zeros((lam.size, B))
    for g, lv in enumerate(lam):
        ks[g], _ = optimal_stopping(p, Cc, float(lv), risk_adjustment, alpha, beta)
        for b in range(B):
            pb, cs = 1.0, 0.0
            for i in range(int(ks[g, b]) + 1):
                pb *= p[b, i]
                cs += Cc[i]
            cost[g, b], ok[g, b] = cs, pb
    return ks, cost, ok


def commit_step(tok, n_acc, drawn, seq_len, out_tokens, max_len: Optional[int] = None):
    """N3 oracle (numpy, integer): returns (new seq_len, new out_tokens, n_commit); inputs are not modified."""
    tok = np.asarray(tok, dtype=np.int32)
    B, K = tok.shape
    out = np.array(out_tokens, dtype=np.int32, copy=True)
    lens = np.array(seq_len, dtype=np.int32, copy=True)
    cap = out.shape[1] if max_len is None else int(max_len)
    n_commit = np.zeros(B, np.int32)
    for b in range(B):
        na = int(min(max(int(n_acc[b]), 0), K))
        new = list(tok[b, :na]) + [int(drawn[b])]
        room = max(0, cap - int(lens[b]))
        new = new[:room]
        out[b, lens[b]:lens[b] + len(new)] = new
        n_commit[b] = len(new)
        lens[b] += len(new)
    return lens, out, n_commit


# ----------------------------------------------------------------------------- A7
def logprob_stats(lp, n_valid=None, K: Optional[int] = None) -> np.ndarray:
    lib = _load()
    lp = _c(lp, np.float32)
    if lp.ndim == 1:
        lp = lp[None, :]
    B, ld = lp.shape
    if K is None:
        K = ld
    nv = None if n_valid is None else _c(n_valid, np.int32)
    out = np.empty((B, 5), np.float64)
    lib.oracle_logprob_stats(_p(lp), C.c_int64(ld), _p(nv), C.c_int(B), C.c_int(K), _p(out))
    return out


def py_logprob_stats(logprobs: Sequence[float]) -> List[float]:
    """generate_training_data.py:166-175 verbatim in behaviour (numpy does the arithmetic)."""
    if len(logprobs):
        return [float(np.mean(logprobs)), float(np.std(logprobs)), float(np.min(logprobs)),
                float(np.percentile(logprobs, 25)), float(np.median(logprobs))]
    return [0.0] * 5


# ----------------------------------------------------------------------------- A8 / A11
def mlp_predict(x, w1, b1, w2, b2) -> np.ndarray:
    lib = _load()
    x = _c(x, np.float32)
    if x.ndim == 1:
        x = x[None, :]
    w1 = _c(w1, np.float32)
    hidden, in_dim = w1.shape
    b1 = _c(b1, np.float32).reshape(-1)
    w2 = _c(w2, np.float32).reshape(-1)
    b2 = _c(b2, np.float32).reshape(-1)
    B = x.shape[0]
    out = np.empty(B, np.float32)
    lib.oracle_mlp_predict(_p(x), C.c_int64(x.shape[1]), _p(w1), _p(b1), _p(w2), _p(b2), C.c_int(B),
                           C.c_int(in_dim), C.c_int(hidden), _p(out))
    return out


def threshold_stop(score, theta) -> np.ndarray:
    lib = _load()
    score = _c(score, np.float32).reshape(-1)
    theta = _c(theta, np.float64).reshape(-1)
    out = np.empty(score.size, np.int32)
    lib.oracle_threshold_stop(_p(score), _p(theta), C.c_int(score.size), C.c_int(theta.size), _p(out))
    return out


# ----------------------------------------------------------------------------- A1 / A2 / A3 / A10
def bayes_adjust(p, n_obs: int, alpha: float = 1.0, beta: float = 1.0) -> np.ndarray:
    lib = _load()
    p = _c(p, np.float64).reshape(-1)
    out = np.empty_like(p)
    lib.oracle_bayes_adjust(_p(p), C.c_int64(n_obs), C.c_double(alpha), C.c_double(beta),
                            C.c_int(p.size), _p(out))
    return out


def optimal_stopping(p, Cc, lam: float, risk_adjustment: bool = False, alpha: float = 1.0,
                     beta: float = 1.0) -> Tuple[np.ndarray, np.ndarray]:
    lib = _load()
    p = _c(p, np.float64)
    if p.ndim == 1:
        p = p[None, :]
    B, L = p.shape
    Cc = _c(Cc, np.float64).reshape(-1)
    if Cc.size != L:
        raise ValueError("p and C must have the same length")  # dp_solver.py:34-35
    k = np.empty(B, np.int32)
    J = np.empty((B, L + 1), np.float64)
    rc = lib.oracle_optimal_stopping(_p(p), _p(Cc), C.c_double(lam), C.c_int(B), C.c_int(L),
                                     C.c_int(int(risk_adjustment)), C.c_double(alpha),
                                     C.c_double(beta), _p(k), _p(J))
    if rc != 0:
        raise ValueError(rc)
    return k, J


def expected_cost(p, Cc, lam: float, k) -> np.ndarray:
    lib = _load()
    p = _c(p, np.float64)
    if p.ndim == 1:
        p = p[None, :]
    B, L = p.shape
    Cc = _c(Cc, np.float64).reshape(-1)
    k = _c(k, np.int32).reshape(-1)
    out = np.empty(B, np.float64)
    lib.oracle_expected_cost(_p(p), _p(Cc), C.c_double(lam), _p(k), C.c_int(B), C.c_int(L), _p(out))
    return out


def derive_thresholds(q, c, lam: float) -> Tuple[np.ndarray, np.ndarray]:
    lib = _load()
    q = _c(q, np.float64).reshape(-1)
    c = _c(c, np.float64).reshape(-1)
    n = q.size
    theta = np.empty(n, np.float64)
    V = np.empty(n + 1, np.float64)
    rc = lib.oracle_derive_thresholds(_p(q), _p(c), C.c_int(n), C.c_double(lam), _p(theta), _p(V))
    if rc != 0:
        raise ValueError(rc)
    return theta, V


# pure-Python restatements (CPython float == IEEE f64, no FMA), used to cross-check the C code
def py_bayesian_adjustment(p_hat: float, n_obs: int, alpha: float = 1.0, beta: float = 1.0) -> float:
    """dp_solver.py:106-130."""
    pa = n_obs * p_hat + alpha
    pb = n_obs * (1 - p_hat) + beta
    return pa / (pa + pb)


def py_optimal_stopping_rule(p: Sequence[float], Cc: Sequence[float], lam: float,
                             risk_adjustment: bool = False, alpha: float = 1.0,
                             beta: float = 1.0) -> Tuple[int, List[float]]:
    """dp_solver.py:12-71."""
    if len(p) != len(Cc):
        raise ValueError("p and C must have the same length")
    L = len(Cc)
    if risk_adjustment:
        p = [py_bayesian_adjustment(pi, 100, alpha, beta) for pi in p]
    p_bar = [1.0]
    for i in range(L):
        p_bar.append(p_bar[-1] * p[i])
    J = [0.0] * (L + 1)
    stop = [False] * L
    for i in reversed(range(L)):
        a = Cc[i] + lam * (1 - p_bar[i + 1])
        b = Cc[i] + J[i + 1]
        if a <= b:
            stop[i], J[i] = True, a
        else:
            stop[i], J[i] = False, b
    k = next((i for i, s in enumerate(stop) if s), L - 1)
    return k, J


def py_derive_optimal_policy(q: Sequence[float], c: Sequence[float], lam: float) -> List[float]:
    """optimal_stopping.py:45-82."""
    n = len(q)
    V = [0.0] * (n + 1)
    th = [0.0] * n
    for s in range(n - 1, -1, -1):
        r_stop = q[s] - lam * c[s]
        if s < n - 1:
            pi = 0.6 * (1 - q[s])
            r_cont = pi * V[s + 1] + (1 - pi) * r_stop
        else:
            r_cont = -math.inf
        V[s] = max(r_stop, r_cont)
        th[s] = (V[s + 1] + lam * c[s]) / (1 + lam * (c[s + 1] - c[s])) if s < n - 1 else 0.0
    return th
