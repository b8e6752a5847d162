"""Device-tensor front end of libasd_hip.so: torch CUDA tensors in, torch CUDA tensors out.

torch is plumbing here (device memory + the current HIP stream); every function below is one
call through the C ABI with raw device pointers.  Nothing synchronises; outputs are valid in
stream order.  All of them raise if the tensors are not on a GPU -- there is no CPU path.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional, Tuple

import numpy as np
import torch

from . import _binding as B

_DTYPE_CODE = {torch.float32: B.DTYPE_F32, torch.bfloat16: B.DTYPE_BF16, torch.float16: B.DTYPE_F16}


def _lib():
    return B.load_library()


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


# test infrastructure: `with kernels.test_hooks() as lib:` routes THIS thread's calls through lib/libasd_hip_test.so (the
# -DASD_TEST_HOOKS build, which alone has the asd_debug_* switches; `lib` is its ctypes handle)
test_hooks = B.use_test_library


def _dev(t: torch.Tensor, name: str, dtype=None) -> int:
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise ValueError(f"{name} must be a CUDA (HIP) tensor; this package has no CPU path")
    if dtype is not None and t.dtype != dtype:
        raise ValueError(f"{name} must be {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous")
    return t.data_ptr()


def _opt(t: Optional[torch.Tensor], name: str, dtype=None) -> Optional[int]:
    return None if t is None else _dev(t, name, dtype)


def version() -> int:
    return _lib().asd_version()


def device_cu_count(device: int = 0) -> int:
    n = _lib().asd_device_cu_count(device)
    if n < 0:
        B.check("asd_device_cu_count", n)
    return n


# ------------------------------------------------------------------------------- verify + accept
class LostHandoffError(RuntimeError):
    """A kernel's bounded wait for a hand-off word of this workspace ran out (ASD_WS_LOST_HANDOFF): the rows / sequences that
    depended on it came back POISONED (NaN / reject / tok = -1), and the workspace has been re-initialised."""


class _StatusWorkspace:
    """What every hand-off workspace shares (include/asd_hip.h, asd_workspace_status): `buf`, whose first 32-bit word is the
    sticky status the kernels or into when a bounded poll gives up."""

    buf: torch.Tensor
    bytes: int

    def reset(self) -> None:
        B.check("asd_workspace_init", _lib().asd_workspace_init(self.buf.data_ptr(), self.bytes, _stream()))

    @property
    def status_word(self) -> torch.Tensor:
        """0-d int32 view of the status word: read it together with whatever the caller synchronises on anyway."""
        return self.buf[:4].view(torch.int32)[0]

    def status(self) -> int:
        """Synchronising read of the status word through the C ABI (asd_workspace_status)."""
        import ctypes as C
        out = C.c_uint32(0)
        B.check("asd_workspace_status", _lib().asd_workspace_status(self.buf.data_ptr(), C.addressof(out), _stream()))
        return int(out.value)

    def check(self) -> None:
        """Raise LostHandoffError (after re-initialising the workspace) if a hand-off was lost since the last reset."""
        st = self.status()
        if st != 0:
            self.reset()
            raise LostHandoffError(f"{type(self).__name__}: status 0x{st:x} (a hand-off word never arrived; results poisoned, "
                                   "workspace re-initialised)")


class VerifyWorkspace(_StatusWorkspace):
    """Ticket + granule scratch of asd_verify_accept, zeroed once (asd_workspace_init).

    One workspace serves any number of stream-ordered calls with B' <= B, K' <= K; calls that may
    overlap on different streams need one workspace each."""

    def __init__(self, B_: int, K: int, V: int, dtype: torch.dtype = torch.bfloat16,
                 device: Optional[torch.device] = None):
        self.B, self.K, self.V = B_, K, V
        self.dtype = dtype
        self.bytes = int(_lib().asd_verify_accept_workspace_bytes(B_, K, V, _DTYPE_CODE[dtype]))
        self.buf = torch.empty(self.bytes, dtype=torch.uint8, device=device or torch.device("cuda"))
        self.reset()

    def fits(self, B_: int, K: int) -> bool:
        need = int(_lib().asd_verify_accept_workspace_bytes(B_, K, self.V, _DTYPE_CODE[self.dtype]))
        return need <= self.bytes


@dataclass
class VerifyResult:
    lp_target: torch.Tensor   # [B,K] f32   log p_target(tok)
    accept: torch.Tensor      # [B,K] u8
    n_acc: torch.Tensor       # [B]   i32   accepted-prefix length
    accept_bits: torch.Tensor  # [B]  i64 (bit k = accept[b,k])


def _logits_2d(logits: torch.Tensor, Bv: int, K: int) -> Tuple[int, int, int]:
    if logits.dtype not in _DTYPE_CODE:
        raise ValueError(f"logits dtype {logits.dtype} unsupported (f32 / bf16 / f16)")
    if not logits.is_cuda:
        raise ValueError("logits must be a CUDA (HIP) tensor; this package has no CPU path")
    if logits.dim() == 3:
        if logits.shape[0] != Bv or logits.shape[1] != K:
            raise ValueError("logits must be [B,K,V]")
        # the stride of a size-1 dimension is arbitrary (a [1, 64, V] slice of [1, 70, V] keeps stride(0) = 70 V)
        if logits.stride(2) != 1 or (Bv > 1 and logits.stride(0) != K * logits.stride(1)):
            raise ValueError("logits rows must be unit-stride in V and evenly spaced over (b,k)")
        ld = logits.stride(1) if K > 1 else (logits.stride(0) if Bv > 1 else logits.shape[2])
        return logits.shape[2], ld, logits.data_ptr()
    if logits.dim() == 2:
        if logits.shape[0] != Bv * K or logits.stride(1) != 1:
            raise ValueError("2-D logits must be [B*K, V] with unit stride in V")
        return logits.shape[1], logits.stride(0), logits.data_ptr()
    raise ValueError("logits must be [B,K,V] or [B*K,V]")


def verify_accept(logits: torch.Tensor, tok: torch.Tensor, lp_draft: torch.Tensor, u: torch.Tensor,
                  workspace: VerifyWorkspace, out: Optional[VerifyResult] = None, *, inv_temperature: float = 1.0,
                  splits: int = 0, threads: int = 0, unroll: int = 0, nontemporal: int = -1) -> VerifyResult:
    """A5/A6 in one launch (include/asd_hip.h: asd_verify_accept_ex).  tok/lp_draft/u: [B,K].
    inv_temperature scales the logits inside the kernel (the test runs on softmax(logits / T))."""
    Bv, K = tok.shape
    V, ld, ptr = _logits_2d(logits, Bv, K)
    dev = logits.device
    if out is None:
        out = VerifyResult(torch.empty((Bv, K), dtype=torch.float32, device=dev),
                           torch.empty((Bv, K), dtype=torch.uint8, device=dev),
                           torch.empty((Bv,), dtype=torch.int32, device=dev),
                           torch.empty((Bv,), dtype=torch.int64, device=dev))
    opt = B.verify_options(inv_temperature, splits, threads, unroll, nontemporal)
    rc = _lib().asd_verify_accept_ex(
        ptr, _DTYPE_CODE[logits.dtype], ld, _dev(tok, "tok", torch.int32), _dev(lp_draft, "lp_draft", torch.float32),
        _dev(u, "u", torch.float32), Bv, K, V, _dev(out.lp_target, "lp_target", torch.float32),
        _dev(out.accept, "accept", torch.uint8), _dev(out.n_acc, "n_acc", torch.int32),
        _dev(out.accept_bits, "accept_bits", torch.int64), workspace.buf.data_ptr(), workspace.bytes,
        C.addressof(opt), _stream())
    B.check("asd_verify_accept_ex", rc)
    return out


def verify_accept_stats(logits: torch.Tensor, tok: torch.Tensor, lp_draft: torch.Tensor, u: torch.Tensor,
                        workspace: VerifyWorkspace, out: Optional[VerifyResult] = None, *, inv_temperature: float = 1.0,
                        want_entropy: bool = True) -> Tuple[VerifyResult, torch.Tensor, Optional[torch.Tensor]]:
    """asd_verify_accept_stats: the verify pass + per position max log-prob [B,K] and (optionally) the entropy of the
    target softmax [B,K] -- the device-side inputs of the doc-only FeatureExtractor (RESEARCH_PROTOCOL.md:378-400)."""
    Bv, K = tok.shape
    V, ld, ptr = _logits_2d(logits, Bv, K)
    dev = logits.device
    if out is None:
        out = VerifyResult(torch.empty((Bv, K), dtype=torch.float32, device=dev),
                           torch.empty((Bv, K), dtype=torch.uint8, device=dev),
                           torch.empty((Bv,), dtype=torch.int32, device=dev),
                           torch.empty((Bv,), dtype=torch.int64, device=dev))
    max_lp = torch.empty((Bv, K), dtype=torch.float32, device=dev)
    ent = torch.empty((Bv, K), dtype=torch.float32, device=dev) if want_entropy else None
    rc = _lib().asd_verify_accept_stats(
        ptr, _DTYPE_CODE[logits.dtype], ld, _dev(tok, "tok", torch.int32), _dev(lp_draft, "lp_draft", torch.float32),
        _dev(u, "u", torch.float32), Bv, K, V, out.lp_target.data_ptr(), out.accept.data_ptr(), out.n_acc.data_ptr(),
        out.accept_bits.data_ptr(), max_lp.data_ptr(), None if ent is None else ent.data_ptr(), workspace.buf.data_ptr(),
        workspace.bytes, float(inv_temperature), _stream())
    B.check("asd_verify_accept_stats", rc)
    return out, max_lp, ent


def lse_partial(logits_shard: torch.Tensor, tok: torch.Tensor, v_offset: int, workspace: VerifyWorkspace,
                out: Optional[torch.Tensor] = None, inv_temperature: float = 1.0) -> torch.Tensor:
    """Per-shard (m2, s, g) triples, [B,K,3] f32, with sum_v exp(x) = s * 2^m2 (log2 domain)."""
    Bv, K = tok.shape
    V, ld, ptr = _logits_2d(logits_shard, Bv, K)
    if out is None:
        out = torch.empty((Bv, K, 3), dtype=torch.float32, device=logits_shard.device)
    rc = _lib().asd_lse_partial(ptr, _DTYPE_CODE[logits_shard.dtype], ld, _dev(tok, "tok", torch.int32), Bv, K, V,
                                int(v_offset), float(inv_temperature), _dev(out, "msg", torch.float32),
                                workspace.buf.data_ptr(), workspace.bytes, _stream())
    B.check("asd_lse_partial", rc)
    return out


def accept_from_partials(msg_all: torch.Tensor, lp_draft: torch.Tensor, u: torch.Tensor,
                         out: Optional[VerifyResult] = None, inv_temperature: float = 1.0) -> VerifyResult:
    """msg_all: [n_shards,B,K,3] (all-gathered asd_lse_partial outputs, shard order fixed)."""
    n_shards, Bv, K, three = msg_all.shape
    assert three == 3
    dev = msg_all.device
    if out is None:
        out = VerifyResult(torch.empty((Bv, K), dtype=torch.float32, device=dev),
                           torch.empty((Bv, K), dtype=torch.uint8, device=dev),
                           torch.empty((Bv,), dtype=torch.int32, device=dev),
                           torch.empty((Bv,), dtype=torch.int64, device=dev))
    rc = _lib().asd_accept_from_partials(_dev(msg_all, "msg_all", torch.float32), n_shards,
                                         _dev(lp_draft, "lp_draft", torch.float32), _dev(u, "u", torch.float32), Bv, K,
                                         float(inv_temperature), out.lp_target.data_ptr(), out.accept.data_ptr(), out.n_acc.data_ptr(),
                                         out.accept_bits.data_ptr(), _stream())
    B.check("asd_accept_from_partials", rc)
    return out


class LmHeadVerifier:
    """N2: lm_head projection fused with the verify pass (asd_lm_head_verify).  Holds the partials
    workspace for one (B, K, V); `weight` is the [V, D] bf16 or f16 lm_head matrix (nn.Linear layout); the hidden
    states must have the same element type."""

    def __init__(self, weight: torch.Tensor, B_: int, K: int, packed: bool = False, packed_image: Optional[torch.Tensor] = None):
        """Sized for batches of up to B_ sequences of exactly K positions (calls with fewer sequences reuse the workspace).
        packed=True: keep a tile-major copy of the matrix (asd_lm_head_pack_weights; +V*D*2 bytes, e.g. +2.5 GB for the
        152064 x 8192 head) and stream that: every 64-deep reduction step of a column block is one contiguous 32 KiB run.
        packed_image: an image another verifier of the SAME matrix already built (shared, read-only).
        Results are bit-identical either way.  The image is a snapshot: repack after changing the weights."""
        if weight.dim() != 2 or weight.dtype not in (torch.bfloat16, torch.float16) or not weight.is_cuda:
            raise ValueError("weight must be a [V, D] bf16 or f16 CUDA tensor")
        self._dt = _DTYPE_CODE[weight.dtype]
        if weight.stride(1) != 1:
            raise ValueError("weight rows must be contiguous")
        self.weight = weight
        self.B, self.K = int(B_), int(K)
        self.V, self.D = weight.shape
        self._w_ptr, self._ld_w = weight.data_ptr(), weight.stride(0)
        self.packed = None
        if packed_image is not None:
            self.packed = packed_image
            self._w_ptr, self._ld_w = packed_image.data_ptr(), 0
        elif packed:
            nbytes = int(_lib().asd_lm_head_packed_bytes(self.V, self.D))
            if nbytes == 0:
                raise ValueError("D must be a multiple of 64 to pack the lm_head")
            self.packed = torch.empty(nbytes, dtype=torch.uint8, device=weight.device)
            B.check("asd_lm_head_pack_weights", _lib().asd_lm_head_pack_weights(
                weight.data_ptr(), weight.stride(0), self._dt, self.V, self.D, self.packed.data_ptr(), nbytes, _stream()))
            self._w_ptr, self._ld_w = self.packed.data_ptr(), 0
        n = int(_lib().asd_lm_head_verify_workspace_bytes(self.B, self.K, self.V))
        self.workspace = torch.empty(max(n, 256), dtype=torch.uint8, device=weight.device)

    def __call__(self, hidden: torch.Tensor, tok: torch.Tensor, lp_draft: Optional[torch.Tensor] = None,
                 u: Optional[torch.Tensor] = None, out: Optional[VerifyResult] = None, inv_temperature: float = 1.0,
                 greedy: bool = False, argmax_out: Optional[torch.Tensor] = None) -> VerifyResult:
        """hidden [B, K, D] (or [B*K, D]) bf16: the final-norm output the lm_head would consume.
        greedy=True: accept[b,k] = (tok[b,k] == argmax logits[b,k]) (lp_draft / u unused);
        argmax_out: optional [B, K] int32 tensor that receives the row arg-max either way."""
        Bv, K = tok.shape
        if Bv > self.B or K != self.K:
            raise ValueError(f"verifier was sized for B<={self.B}, K={self.K}, got {Bv}, {K}")
        if not greedy and (lp_draft is None or u is None):
            raise ValueError("lp_draft and u are required unless greedy=True")
        h2 = hidden.reshape(Bv * K, hidden.shape[-1]) if hidden.dim() == 3 else hidden
        if h2.dtype != self.weight.dtype or h2.shape != (Bv * K, self.D) or h2.stride(1) != 1:
            raise ValueError("hidden must be [B*K, D] of the weight's element type with contiguous rows")
        dev = h2.device
        if out is None:
            out = VerifyResult(torch.empty((Bv, K), dtype=torch.float32, device=dev),
                               torch.empty((Bv, K), dtype=torch.uint8, device=dev),
                               torch.empty((Bv,), dtype=torch.int32, device=dev),
                               torch.empty((Bv,), dtype=torch.int64, device=dev))
        rc = _lib().asd_lm_head_verify_ex(h2.data_ptr(), h2.stride(0) if Bv * K > 1 else self.D, self._w_ptr,
                                          self._ld_w, self._dt, self.D, _dev(tok, "tok", torch.int32),
                                          _opt(lp_draft, "lp_draft", torch.float32), _opt(u, "u", torch.float32), Bv, K,
                                          self.V, float(inv_temperature), 1 if greedy else 0, out.lp_target.data_ptr(),
                                          out.accept.data_ptr(), out.n_acc.data_ptr(), out.accept_bits.data_ptr(),
                                          _opt(argmax_out, "argmax_out", torch.int32), self.workspace.data_ptr(),
                                          self.workspace.numel(), _stream())
        B.check("asd_lm_head_verify_ex", rc)
        return out


    def partial(self, hidden: torch.Tensor, tok: torch.Tensor, v_offset: int, inv_temperature: float = 1.0,
                msg: Optional[torch.Tensor] = None) -> torch.Tensor:
        """`self.weight` is this rank's vocabulary shard starting at global id `v_offset`: returns the
        [B, K, 3] (m2, s, g) message of asd_lse_partial without forming the shard's logits (asd_lm_head_partial)."""
        Bv, K = tok.shape
        if Bv > self.B or K != self.K:
            raise ValueError(f"verifier was sized for B<={self.B}, K={self.K}, got {Bv}, {K}")
        h2 = hidden.reshape(Bv * K, hidden.shape[-1]) if hidden.dim() == 3 else hidden
        if h2.dtype != self.weight.dtype or h2.shape != (Bv * K, self.D) or h2.stride(1) != 1:
            raise ValueError("hidden must be [B*K, D] of the weight's element type with contiguous rows")
        if msg is None:
            msg = torch.empty((Bv, K, 3), dtype=torch.float32, device=h2.device)
        rc = _lib().asd_lm_head_partial(h2.data_ptr(), h2.stride(0) if Bv * K > 1 else self.D, self._w_ptr,
                                        self._ld_w, self._dt, self.D, _dev(tok, "tok", torch.int32), Bv, K,
                                        self.V, int(v_offset), float(inv_temperature), _dev(msg, "msg", torch.float32),
                                        self.workspace.data_ptr(), self.workspace.numel(), _stream())
        B.check("asd_lm_head_partial", rc)
        return msg


class LinearWorkspace:
    """Slab buffer of asd_linear's reduction slices; grows to the largest (M, N, D) it has served.  One per stream:
    consecutive calls on a stream may share it."""

    def __init__(self, device, on_grow=None):
        """on_grow: called BEFORE the buffer is replaced by a larger one -- whoever captured hipGraphs that hold the old
        buffer's address (SyntheticLM.enable_graphs) must drop them."""
        self.device = torch.device(device)
        self.buf: Optional[torch.Tensor] = None
        self.on_grow = on_grow

    def ensure(self, M: int, N: int, D: int) -> Tuple[int, int]:
        need = int(_lib().asd_linear_workspace_bytes(M, N, D))
        if self.buf is None or self.buf.numel() < need:
            if self.buf is not None and self.on_grow is not None:
                self.on_grow()
            self.buf = torch.empty(max(need, 1 << 20), dtype=torch.uint8, device=self.device)
        return self.buf.data_ptr(), self.buf.numel()


def linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, *, workspace: LinearWorkspace,
           out: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None) -> torch.Tensor:
    """X3: y = x @ weight.T (+ bias) (+ residual) through asd_linear_ex.  x [..., D], weight [N, D] (nn.Linear layout), all bf16
    or all f16 CUDA tensors; returns [..., N] of the same type.  residual [M, N] may be `out` itself (in-place accumulate)."""
    if weight.dim() != 2 or weight.dtype not in (torch.bfloat16, torch.float16) or not weight.is_cuda or weight.stride(1) != 1:
        raise ValueError("weight must be a [N, D] bf16 or f16 CUDA tensor with contiguous rows")
    if x.dtype != weight.dtype or not x.is_cuda or x.shape[-1] != weight.shape[1]:
        raise ValueError("x must be a CUDA tensor of the weight's element type with D trailing elements")
    N, D = weight.shape
    x2 = x.reshape(-1, D)
    if x2.stride(1) != 1:
        x2 = x2.contiguous()
    M = x2.shape[0]
    if out is None:
        out = torch.empty((M, N), dtype=x.dtype, device=x.device)
    elif out.shape != (M, N) or out.dtype != x.dtype or out.stride(1) != 1:
        raise ValueError("out must be a [M, N] tensor of the operands' type with contiguous rows")
    if bias is not None and (bias.dtype != x.dtype or bias.numel() != N or not bias.is_contiguous()):
        raise ValueError("bias must be a contiguous [N] tensor of the operands' type")
    if residual is not None and (residual.shape != (M, N) or residual.dtype != x.dtype or residual.stride(1) != 1):
        raise ValueError("residual must be a [M, N] tensor of the operands' type with contiguous rows")
    ws_ptr, ws_bytes = workspace.ensure(M, N, D)
    rc = _lib().asd_linear_ex(x2.data_ptr(), x2.stride(0), weight.data_ptr(), weight.stride(0),
                              None if bias is None else bias.data_ptr(), None if residual is None else residual.data_ptr(),
                              0 if residual is None else residual.stride(0), _DTYPE_CODE[x.dtype], M, N, D, out.data_ptr(),
                              out.stride(0), ws_ptr, ws_bytes, _stream())
    B.check("asd_linear_ex", rc)
    return out.view(*x.shape[:-1], N)


def commit_step(tok: torch.Tensor, n_acc: torch.Tensor, drawn: torch.Tensor, seq_len: torch.Tensor,
                out_tokens: torch.Tensor, n_commit: Optional[torch.Tensor] = None, max_len: Optional[int] = None) -> None:
    """N3: append every sequence's accepted prefix + drawn token to its row of `out_tokens` and advance
    `seq_len` in place (asd_commit_step).  tok [B,K] i32, n_acc / drawn / seq_len [B] i32, out_tokens [B, T] i32."""
    Bv, K = tok.shape
    if out_tokens.dim() != 2 or out_tokens.shape[0] != Bv or out_tokens.stride(1) != 1:
        raise ValueError("out_tokens must be [B, T] int32 with contiguous rows")
    cap = out_tokens.shape[1] if max_len is None else int(max_len)
    rc = _lib().asd_commit_step(_dev(tok, "tok", torch.int32), _dev(n_acc, "n_acc", torch.int32),
                                _dev(drawn, "drawn", torch.int32), Bv, K, _dev(seq_len, "seq_len", torch.int32),
                                _dev(out_tokens, "out_tokens", torch.int32), out_tokens.stride(0),
                                _opt(n_commit, "n_commit", torch.int32), cap, _stream())
    B.check("asd_commit_step", rc)


# ------------------------------------------------------------------------------- predictor side
def logprob_stats(lp: torch.Tensor, n_valid: Optional[torch.Tensor] = None) -> torch.Tensor:
    """A7: [B,K] f32 log-probs -> [B,5] f64 (mean, std, min, q25, median), numpy semantics."""
    Bv, K = lp.shape
    out = torch.empty((Bv, 5), dtype=torch.float64, device=lp.device)
    rc = _lib().asd_logprob_stats(_dev(lp, "lp", torch.float32), lp.stride(0) if Bv else K,
                                  _opt(n_valid, "n_valid", torch.int32), Bv, K, out.data_ptr(), _stream())
    B.check("asd_logprob_stats", rc)
    return out


def pack_mlp_weights(w1, b1, w2, b2, device=None) -> torch.Tensor:
    """state_dict tensors (net.0.weight [H,D], net.0.bias [H], net.3.weight [1,H], net.3.bias [1])
    -> packed device buffer of asd_mlp_predict."""
    w1 = np.ascontiguousarray(torch.as_tensor(w1).detach().cpu().numpy(), dtype=np.float32)
    H, D = w1.shape
    b1 = np.ascontiguousarray(torch.as_tensor(b1).detach().cpu().numpy(), dtype=np.float32).reshape(H)
    w2 = np.ascontiguousarray(torch.as_tensor(w2).detach().cpu().numpy(), dtype=np.float32).reshape(H)
    b2 = np.ascontiguousarray(torch.as_tensor(b2).detach().cpu().numpy(), dtype=np.float32).reshape(1)
    n = int(_lib().asd_mlp_packed_floats(D, H))
    packed = np.empty(n, np.float32)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    B.check("asd_mlp_pack_weights", _lib().asd_mlp_pack_weights(vp(w1), vp(b1), vp(w2), vp(b2), D, H, vp(packed)))
    t = torch.from_numpy(packed)
    return t.to(device or torch.device("cuda"))


def mlp_predict(x: torch.Tensor, packed_w: torch.Tensor, in_dim: int, hidden: int) -> torch.Tensor:
    """A8 (eval mode): x [B,in_dim] f32 -> score [B] f32."""
    Bv = x.shape[0]
    if not x.is_cuda or x.dtype != torch.float32 or x.dim() != 2 or (Bv and x.stride(1) != 1):
        raise ValueError("x must be a [B, in_dim] float32 CUDA tensor with unit stride along in_dim (rows may be strided)")
    out = torch.empty((Bv,), dtype=torch.float32, device=x.device)
    rc = _lib().asd_mlp_predict(x.data_ptr(), x.stride(0) if Bv else in_dim,
                                _dev(packed_w, "packed_w", torch.float32), Bv, in_dim, hidden, out.data_ptr(), _stream())
    B.check("asd_mlp_predict", rc)
    return out


def threshold_stop(score: torch.Tensor, theta: torch.Tensor) -> torch.Tensor:
    """A11: first stage s with score >= theta[s] (or the last)."""
    out = torch.empty((score.shape[0],), dtype=torch.int32, device=score.device)
    rc = _lib().asd_threshold_stop(_dev(score, "score", torch.float32), _dev(theta, "theta", torch.float64),
                                   score.shape[0], theta.shape[0], out.data_ptr(), _stream())
    B.check("asd_threshold_stop", rc)
    return out


def bayes_adjust(p: torch.Tensor, n_obs: int, alpha: float = 1.0, beta: float = 1.0) -> torch.Tensor:
    """A2 over a flat f64 tensor."""
    out = torch.empty_like(p)
    rc = _lib().asd_bayes_adjust(_dev(p, "p", torch.float64), int(n_obs), float(alpha), float(beta), p.numel(),
                                 out.data_ptr(), _stream())
    B.check("asd_bayes_adjust", rc)
    return out


def optimal_stopping(p: torch.Tensor, Cc: torch.Tensor, lam: float, risk_adjustment: bool = False,
                     alpha: float = 1.0, beta: float = 1.0, want_J: bool = True):
    """A1 batched: p [B,L] f64, C [L] f64 -> (k_star [B] i32, J [B,L+1] f64 or None)."""
    Bv, L = p.shape
    if Cc.numel() != L:
        raise ValueError("p and C must have the same length")  # dp_solver.py:34-35
    k = torch.empty((Bv,), dtype=torch.int32, device=p.device)
    J = torch.empty((Bv, L + 1), dtype=torch.float64, device=p.device) if want_J else None
    rc = _lib().asd_optimal_stopping(_dev(p, "p", torch.float64), _dev(Cc, "C", torch.float64), float(lam), Bv, L,
                                     int(bool(risk_adjustment)), float(alpha), float(beta), k.data_ptr(),
                                     None if J is None else J.data_ptr(), _stream())
    B.check("asd_optimal_stopping", rc)
    return k, J


def lambda_sweep(p: torch.Tensor, Cc: torch.Tensor, lam: torch.Tensor, risk_adjustment: bool = False, alpha: float = 1.0,
                 beta: float = 1.0) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """N4: p [B,L] f64, Cc [L] f64, lam [G] f64 -> (k_star [G,B] i32, cost [G,B] f64, p_ok [G,B] f64)."""
    Bv, L = p.shape
    G = lam.shape[0]
    k = torch.empty((G, Bv), dtype=torch.int32, device=p.device)
    cost = torch.empty((G, Bv), dtype=torch.float64, device=p.device)
    ok = torch.empty((G, Bv), dtype=torch.float64, device=p.device)
    rc = _lib().asd_lambda_sweep(_dev(p, "p", torch.float64), _dev(Cc, "C", torch.float64), _dev(lam, "lam", torch.float64),
                                 Bv, L, G, 1 if risk_adjustment else 0, float(alpha), float(beta), k.data_ptr(),
                                 cost.data_ptr(), ok.data_ptr(), _stream())
    B.check("asd_lambda_sweep", rc)
    return k, cost, ok


def expected_cost(p: torch.Tensor, Cc: torch.Tensor, lam: float, k: torch.Tensor) -> torch.Tensor:
    """A3 batched."""
    Bv, L = p.shape
    out = torch.empty((Bv,), dtype=torch.float64, device=p.device)
    rc = _lib().asd_expected_cost(_dev(p, "p", torch.float64), _dev(Cc, "C", torch.float64), float(lam),
                                  _dev(k, "k", torch.int32), Bv, L, out.data_ptr(), _stream())
    B.check("asd_expected_cost", rc)
    return out


def derive_thresholds(q, c, lam: float) -> Tuple[np.ndarray, np.ndarray]:
    """A10 (host entry point of the library; O(n) f64, once per set_lambda)."""
    q = np.ascontiguousarray(q, dtype=np.float64).reshape(-1)
    c = np.ascontiguousarray(c, dtype=np.float64).reshape(-1)
    if q.size != c.size:
        raise ValueError("quality_bounds and cost_ratios must have the same length")
    theta = np.empty(q.size, np.float64)
    V = np.empty(q.size + 1, np.float64)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    B.check("asd_derive_thresholds", _lib().asd_derive_thresholds(vp(q), vp(c), q.size, float(lam), vp(theta), vp(V)))
    return theta, V


@dataclass
class StopResult:
    score: torch.Tensor       # [B] f32   predictor output
    k_star: Optional[torch.Tensor]    # [B] i32   DP rule
    stop: Optional[torch.Tensor]      # [B] u8    k_star == stage_idx
    thr_stop: Optional[torch.Tensor]  # [B] u8    score >= theta[stage_idx] (or last stage)
    stats: Optional[torch.Tensor]     # [B,5] f64


def predictor_stop(feat: torch.Tensor, packed_w: torch.Tensor, in_dim: int, hidden: int, *, stage_idx: int, L: int,
                   lp: Optional[torch.Tensor] = None, n_valid: Optional[torch.Tensor] = None, stats_col: int = -1,
                   risk_adjustment: bool = True, n_obs: int = 100, alpha: float = 1.0, beta: float = 1.0,
                   p_hist: Optional[torch.Tensor] = None, Cc: Optional[torch.Tensor] = None, lam: float = 1.0,
                   prefix_rule: bool = False, theta: Optional[torch.Tensor] = None,
                   want_stats: bool = False) -> StopResult:
    """Fused post-verify epilogue (asd_predictor_stop): stats -> features -> MLP -> Bayes -> DP / theta."""
    Bv = feat.shape[0]
    dev = feat.device
    score = torch.empty((Bv,), dtype=torch.float32, device=dev)
    dp = p_hist is not None and Cc is not None
    k_star = torch.empty((Bv,), dtype=torch.int32, device=dev) if dp else None
    stop = torch.empty((Bv,), dtype=torch.uint8, device=dev) if dp else None
    thr = torch.empty((Bv,), dtype=torch.uint8, device=dev) if theta is not None else None
    stats = torch.empty((Bv, 5), dtype=torch.float64, device=dev) if want_stats else None
    K = 0 if lp is None else lp.shape[1]
    rc = _lib().asd_predictor_stop(
        _opt(lp, "lp", torch.float32), (lp.stride(0) if lp is not None and Bv else K), _opt(n_valid, "n_valid", torch.int32),
        K, _dev(feat, "feat", torch.float32), feat.stride(0) if Bv else in_dim, stats_col,
        _dev(packed_w, "packed_w", torch.float32), in_dim, hidden, int(bool(risk_adjustment)), int(n_obs), float(alpha),
        float(beta), _opt(p_hist, "p_hist", torch.float64), _opt(Cc, "C", torch.float64), float(lam), L, stage_idx,
        int(bool(prefix_rule)), _opt(theta, "theta", torch.float64), Bv, score.data_ptr(),
        None if k_star is None else k_star.data_ptr(), None if stop is None else stop.data_ptr(),
        None if thr is None else thr.data_ptr(), None if stats is None else stats.data_ptr(), _stream())
    B.check("asd_predictor_stop", rc)
    return StopResult(score, k_star, stop, thr, stats)


def verify_accept_fused(logits: torch.Tensor, tok: torch.Tensor, lp_draft: torch.Tensor, u: torch.Tensor,
                        workspace: VerifyWorkspace, feat: torch.Tensor, packed_w: torch.Tensor, in_dim: int, hidden: int, *,
                        stage_idx: int, L: int, stats_col: int = 5, risk_adjustment: bool = True, n_obs: int = 100,
                        alpha: float = 1.0, beta: float = 1.0, p_hist: Optional[torch.Tensor] = None,
                        Cc: Optional[torch.Tensor] = None, lam: float = 1.0, prefix_rule: bool = False,
                        theta: Optional[torch.Tensor] = None, want_stats: bool = False,
                        out: Optional[VerifyResult] = None, inv_temperature: float = 1.0) -> Tuple[VerifyResult, StopResult]:
    """N1, second form: verify + accept + predictor/stop epilogue in ONE call (one launch for the reference's
    64->32->1 predictor at any batch size, two for other predictor shapes; asd_verify_accept_fused_ex in include/asd_hip.h)."""
    Bv, K = tok.shape
    V, ld, ptr = _logits_2d(logits, Bv, K)
    dev = logits.device
    if out is None:
        out = VerifyResult(torch.empty((Bv, K), dtype=torch.float32, device=dev),
                           torch.empty((Bv, K), dtype=torch.uint8, device=dev),
                           torch.empty((Bv,), dtype=torch.int32, device=dev),
                           torch.empty((Bv,), dtype=torch.int64, device=dev))
    score = torch.empty((Bv,), dtype=torch.float32, device=dev)
    dp = p_hist is not None and Cc is not None
    k_star = torch.empty((Bv,), dtype=torch.int32, device=dev) if dp else None
    stop = torch.empty((Bv,), dtype=torch.uint8, device=dev) if dp else None
    thr = torch.empty((Bv,), dtype=torch.uint8, device=dev) if theta is not None else None
    stats = torch.empty((Bv, 5), dtype=torch.float64, device=dev) if want_stats else None
    opt = B.verify_options(inv_temperature)
    rc = _lib().asd_verify_accept_fused_ex(
        ptr, _DTYPE_CODE[logits.dtype], ld, _dev(tok, "tok", torch.int32), _dev(lp_draft, "lp_draft", torch.float32),
        _dev(u, "u", torch.float32), Bv, K, V, out.lp_target.data_ptr(), out.accept.data_ptr(), out.n_acc.data_ptr(),
        out.accept_bits.data_ptr(), workspace.buf.data_ptr(), workspace.bytes,
        _dev(feat, "feat", torch.float32), feat.stride(0) if Bv else in_dim, stats_col,
        _dev(packed_w, "packed_w", torch.float32), in_dim, hidden, int(bool(risk_adjustment)), int(n_obs), float(alpha),
        float(beta), _opt(p_hist, "p_hist", torch.float64), _opt(Cc, "C", torch.float64), float(lam), L, stage_idx,
        int(bool(prefix_rule)), _opt(theta, "theta", torch.float64), score.data_ptr(),
        None if k_star is None else k_star.data_ptr(), None if stop is None else stop.data_ptr(),
        None if thr is None else thr.data_ptr(), None if stats is None else stats.data_ptr(), C.addressof(opt), _stream())
    B.check("asd_verify_accept_fused_ex", rc)
    return out, StopResult(score, k_star, stop, thr, stats)


def _rows(t: torch.Tensor, name: str) -> Tuple[int, int]:
    """(data_ptr, row stride in elements) of a [rows, V] or [B, K, V] logits tensor with unit stride in V."""
    if not t.is_cuda or t.dtype not in _DTYPE_CODE or t.stride(-1) != 1:
        raise ValueError(f"{name} must be a CUDA f32/bf16/f16 tensor with unit stride along the vocabulary")
    if t.dim() == 3 and t.stride(0) != t.shape[1] * t.stride(1):
        raise ValueError(f"{name}: rows must be evenly spaced over (b, k)")
    return t.data_ptr(), t.stride(-2)


class ResidualSampler(_StatusWorkspace):
    """asd_residual_sample with its workspace: the token each sequence commits after its accepted prefix."""

    def __init__(self, B_: int, V: int, dtype: torch.dtype = torch.bfloat16, device: Optional[torch.device] = None):
        self.B, self.V, self.dtype = B_, V, dtype
        # scratch of the multi-launch form + the mailboxes of the group form (B <= 64): zeroed ONCE, handed back empty by every call
        self.bytes = int(_lib().asd_residual_sample_workspace_bytes(B_, V, _DTYPE_CODE[dtype]))
        self.buf = torch.empty(self.bytes, dtype=torch.uint8, device=device or torch.device("cuda"))
        self.reset()

    def __call__(self, t_logits: torch.Tensor, d_logits: torch.Tensor, n_acc: torch.Tensor, r: torch.Tensor,
                 bonus_logits: Optional[torch.Tensor] = None, inv_temperature: float = 1.0,
                 out: Optional[torch.Tensor] = None, d_threshold: Optional[torch.Tensor] = None) -> torch.Tensor:
        """t_logits / d_logits: [B,K,V]; bonus_logits: [B,V] or None; n_acc: [B] i32; r: [B] f32 -> token [B] i32.
        d_threshold: [B,K] f32 nucleus thresholds of the draft rows (DraftSampler's `thr`) when the drafts were
        drawn with top-p (asd_residual_sample_ex); None = untruncated drafts."""
        Bv, K, V = t_logits.shape
        if d_logits.shape != t_logits.shape or d_logits.dtype != t_logits.dtype:
            raise ValueError("t_logits and d_logits must have the same shape and dtype")
        tp, ldt = _rows(t_logits, "t_logits")
        dp, ldd = _rows(d_logits, "d_logits")
        bp, ldb = (None, V) if bonus_logits is None else _rows(bonus_logits, "bonus_logits")
        if out is None:
            out = torch.empty((Bv,), dtype=torch.int32, device=t_logits.device)
        if d_threshold is not None and tuple(d_threshold.shape) != (Bv, K):
            raise ValueError("d_threshold must be [B, K]")
        rc = _lib().asd_residual_sample_ex(tp, ldt, dp, ldd, bp, ldb, _DTYPE_CODE[t_logits.dtype],
                                           _dev(n_acc, "n_acc", torch.int32), _dev(r, "r", torch.float32), Bv, K, V,
                                           float(inv_temperature), _opt(d_threshold, "d_threshold", torch.float32),
                                           out.data_ptr(), self.buf.data_ptr(), self.bytes, _stream())
        B.check("asd_residual_sample_ex", rc)
        return out


@dataclass
class DraftDraw:
    tok: torch.Tensor   # [B] i32  the proposed token
    lp: torch.Tensor    # [B] f32  log q(tok) under the (nucleus-renormalised) draft distribution
    thr: torch.Tensor   # [B] f32  nucleus threshold logit (-inf: no truncation)


class DraftSampler(_StatusWorkspace):
    """X1: asd_draft_sample with its workspace -- one call proposes the next token of every sequence from the
    draft tier's next-token logits [B, V]: temperature and top-p folded in, inverse-CDF draw from caller-supplied
    uniforms, log q(tok) for the verify step, nucleus threshold for the exact residual."""

    def __init__(self, B_: int, V: int, dtype: torch.dtype = torch.bfloat16, device: Optional[torch.device] = None):
        self.B, self.V, self.dtype = B_, V, dtype
        # mailboxes of the workgroups a row is spread over (B <= 128): zeroed ONCE, handed back empty by every call
        self.bytes = int(_lib().asd_draft_sample_workspace_bytes(B_, V, _DTYPE_CODE[dtype]))
        self.buf = torch.empty(self.bytes, dtype=torch.uint8, device=device or torch.device("cuda"))
        self.reset()

    def __call__(self, logits: torch.Tensor, r: torch.Tensor, inv_temperature: float = 1.0, top_p: float = 1.0,
                 out: Optional[DraftDraw] = None) -> DraftDraw:
        if logits.dim() != 2 or logits.dtype != self.dtype or not logits.is_cuda or logits.stride(1) != 1:
            raise ValueError(f"logits must be a [B, V] {self.dtype} CUDA tensor with unit stride along V")
        Bv, V = logits.shape
        if Bv > self.B or V != self.V:
            raise ValueError(f"sampler was sized for B<={self.B}, V={self.V}")
        dev = logits.device
        if out is None:
            out = DraftDraw(torch.empty((Bv,), dtype=torch.int32, device=dev),
                            torch.empty((Bv,), dtype=torch.float32, device=dev),
                            torch.empty((Bv,), dtype=torch.float32, device=dev))
        rc = _lib().asd_draft_sample(logits.data_ptr(), logits.stride(0) if Bv > 1 else V, _DTYPE_CODE[logits.dtype],
                                     _dev(r, "r", torch.float32), Bv, V, float(inv_temperature), float(top_p),
                                     _dev(out.tok, "tok", torch.int32), _dev(out.lp, "lp", torch.float32),
                                     _dev(out.thr, "thr", torch.float32), self.buf.data_ptr(), self.bytes, _stream())
        B.check("asd_draft_sample", rc)
        return out
