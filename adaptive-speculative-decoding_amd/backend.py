"""Host-level (numpy in / numpy out) view of the kernels, used by the reference-signature
functions (`optimal_stopping_rule(p, C, lam)` with Python lists, `MinimalAdaptiveDecoder.decode`,
`AdaptiveSpeculativePipeline.process_request`).

`HipBackend` is the product: every method uploads, launches libasd_hip.so kernels on the current
HIP device and downloads.  It raises if no GPU / no library is present -- there is no CPU
fallback.  The token-level hot loop does not go through this layer at all; it calls
`kernels.py` on resident device tensors.

`set_backend()` exists so that the CPU test-suite can inject a checker (tests/oracle_backend.py)
to exercise the host control flow of the pipeline without a GPU.  Nothing in this package ever
installs anything but `HipBackend`.
"""
from __future__ import annotations

import threading
from typing import Optional, Tuple

import numpy as np

_backend = None


class HipBackend:
    name = "hip-gfx950"

    def __init__(self, device: Optional[int] = None):
        import torch

        from . import _binding

        _binding.load_library()  # fail loudly when the extension is missing
        if not torch.cuda.is_available():
            raise RuntimeError(
                "adaptive-speculative-decoding_amd needs an AMD GPU (MI355X / gfx950): "
                "torch.cuda.is_available() is False and there is no CPU fallback")
        self._torch = torch
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        # verify workspaces are reused across calls, one set per calling thread (the pipeline serves
        # requests from a thread pool; a workspace must not be shared by calls that may overlap)
        self._tls = threading.local()

    # -- helpers
    def _up(self, a, dtype):
        t = self._torch.from_numpy(np.ascontiguousarray(a, dtype=dtype))
        return t.to(self.device)

    # -- A1
    def optimal_stopping(self, p, Cc, lam, risk_adjustment=False, alpha=1.0, beta=1.0) -> Tuple[np.ndarray, np.ndarray]:
        from . import kernels as K

        p = np.ascontiguousarray(p, dtype=np.float64)
        if p.ndim == 1:
            p = p[None, :]
        Cc = np.ascontiguousarray(Cc, dtype=np.float64).reshape(-1)
        if p.shape[1] != Cc.size:
            raise ValueError("p and C must have the same length")
        k, J = K.optimal_stopping(self._up(p, np.float64), self._up(Cc, np.float64), lam, risk_adjustment, alpha, beta)
        return k.cpu().numpy(), J.cpu().numpy()

    # -- A2
    def bayes_adjust(self, p, n_obs, alpha=1.0, beta=1.0) -> np.ndarray:
        from . import kernels as K

        p = np.ascontiguousarray(p, dtype=np.float64).reshape(-1)
        return K.bayes_adjust(self._up(p, np.float64), n_obs, alpha, beta).cpu().numpy()

    # -- A3
    def expected_cost(self, p, Cc, lam, k) -> np.ndarray:
        from . import kernels as K

        p = np.ascontiguousarray(p, dtype=np.float64)
        if p.ndim == 1:
            p = p[None, :]
        return K.expected_cost(self._up(p, np.float64), self._up(Cc, np.float64), lam,
                               self._up(np.asarray(k).reshape(-1), np.int32)).cpu().numpy()

    # -- N4
    def lambda_sweep(self, p, Cc, lam, risk_adjustment=False, alpha=1.0, beta=1.0):
        from . import kernels as K

        p = np.ascontiguousarray(p, dtype=np.float64)
        Cc = np.ascontiguousarray(Cc, dtype=np.float64).reshape(-1)
        if p.shape[1] != Cc.size:
            raise ValueError("p and C must have the same length")
        k, cost, ok = K.lambda_sweep(self._up(p, np.float64), self._up(Cc, np.float64),
                                     self._up(np.ascontiguousarray(lam, dtype=np.float64).reshape(-1), np.float64),
                                     risk_adjustment, alpha, beta)
        return k.cpu().numpy(), cost.cpu().numpy(), ok.cpu().numpy()

    # -- A10
    def derive_thresholds(self, q, c, lam) -> np.ndarray:
        from . import kernels as K

        return K.derive_thresholds(q, c, lam)[0]

    # -- A8 / A11
    def mlp_predict(self, x, w1, b1, w2, b2) -> np.ndarray:
        from . import kernels as K

        x = np.ascontiguousarray(x, dtype=np.float32)
        if x.ndim == 1:
            x = x[None, :]
        H, D = np.asarray(w1).shape
        packed = K.pack_mlp_weights(w1, b1, w2, b2, device=self.device)
        return K.mlp_predict(self._up(x, np.float32), packed, D, H).cpu().numpy()

    def threshold_stop(self, score, theta) -> np.ndarray:
        from . import kernels as K

        return K.threshold_stop(self._up(np.asarray(score).reshape(-1), np.float32),
                                self._up(np.asarray(theta).reshape(-1), np.float64)).cpu().numpy()

    # -- A7
    def logprob_stats(self, lp, n_valid=None) -> np.ndarray:
        from . import kernels as K

        lp = np.ascontiguousarray(lp, dtype=np.float32)
        if lp.ndim == 1:
            lp = lp[None, :]
        nv = None if n_valid is None else self._up(np.asarray(n_valid).reshape(-1), np.int32)
        return K.logprob_stats(self._up(lp, np.float32), nv).cpu().numpy()

    # -- A5 / A6 (host convenience; the hot loop uses kernels.verify_accept on resident tensors)
    def verify_accept(self, logits, tok, lp_draft, u):
        from . import kernels as K

        torch = self._torch
        lg = logits if isinstance(logits, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(logits))
        lg = lg.to(self.device)
        Bv, Kk, V = lg.shape
        cache = getattr(self._tls, "ws", None)
        if cache is None:
            cache = self._tls.ws = {}
        key = (Bv, Kk, str(lg.dtype))
        ws = cache.get(key)
        if ws is None or ws.V < V:
            ws = cache[key] = K.VerifyWorkspace(Bv, Kk, V, lg.dtype, self.device)
        r = K.verify_accept(lg, self._up(tok, np.int32).reshape(Bv, Kk), self._up(lp_draft, np.float32).reshape(Bv, Kk),
                            self._up(u, np.float32).reshape(Bv, Kk), ws)
        return dict(lp_t=r.lp_target.cpu().numpy(), accept=r.accept.cpu().numpy(), n_acc=r.n_acc.cpu().numpy(),
                    bits=r.accept_bits.cpu().numpy().view(np.uint64))


def get_backend():
    """The process-wide backend; created on first use (=> fails loudly without GPU + library)."""
    global _backend
    if _backend is None:
        _backend = HipBackend()
    return _backend


def set_backend(backend) -> None:
    """Install a backend object (None resets to lazy HipBackend).  Test hook; see module docstring."""
    global _backend
    _backend = backend
