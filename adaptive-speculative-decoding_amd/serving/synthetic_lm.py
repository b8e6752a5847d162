"""Synthetic random-weight Qwen2.5-shape language models -- PLUMBING for the token-level loop.

The reference delegates model execution to vLLM / transformers (third party, parity
unpinned) and fetches checkpoints by name, which is impossible offline.  To feed the verify kernel
real `[B, K, V]` target logits this module builds the same ARCHITECTURE with locally initialised
random weights: RMSNorm, grouped-query attention with rotary embeddings, SwiGLU MLP, untied
lm_head.  Plain PyTorch-ROCm modules (rocBLAS / hipBLASLt GEMMs, SDPA); nothing here is a
hand-written kernel and nothing here is measured by bench.py's headline number.

Shapes (hidden, layers, heads, kv_heads, intermediate) follow the Qwen2.5 model cards; vocab is
152064 for all of them.  `tiny(...)` is BASELINE configs[0]: 2 layers, hidden 128, vocab 1000.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import List, Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F


@dataclass
class LMShape:
    name: str
    hidden: int
    layers: int
    heads: int
    kv_heads: int
    intermediate: int
    vocab: int = 152064
    rope_theta: float = 1.0e6
    rms_eps: float = 1e-6

    @property
    def head_dim(self) -> int:
        return self.hidden // self.heads

    def param_count(self) -> int:
        h, kv = self.hidden, self.kv_heads * self.head_dim
        per_layer = h * h * 2 + h * kv * 2 + 3 * h * self.intermediate + 2 * h
        return self.layers * per_layer + 2 * self.vocab * h + h


QWEN25_SHAPES = {
    "7b": LMShape("qwen2.5-7b", 3584, 28, 28, 4, 18944),
    "14b": LMShape("qwen2.5-14b", 5120, 48, 40, 8, 13824),
    "32b": LMShape("qwen2.5-32b", 5120, 64, 40, 8, 27648),
    "72b": LMShape("qwen2.5-72b", 8192, 80, 64, 8, 29568),
}


def tiny(vocab: int = 1000, hidden: int = 128, layers: int = 2, heads: int = 4, kv_heads: int = 2) -> LMShape:
    return LMShape("tiny", hidden, layers, heads, kv_heads, hidden * 3, vocab, rope_theta=10000.0)


class RMSNorm(nn.Module):
    def __init__(self, dim: int, eps: float, **kw):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(dim, **kw))
        self.eps = eps

    def forward(self, x):
        v = x.float()
        v = v * torch.rsqrt(v.pow(2).mean(-1, keepdim=True) + self.eps)
        return (v * self.weight.float()).to(x.dtype)


def _rope(x: torch.Tensor, pos: torch.Tensor, theta: float) -> torch.Tensor:
    """x: [B, H, T, D]; pos: [T] absolute positions shared by the batch, or [B, T] per sequence."""
    d = x.shape[-1]
    inv = 1.0 / (theta ** (torch.arange(0, d, 2, device=x.device, dtype=torch.float32) / d))
    ang = pos.float()[..., None] * inv
    cos, sin = (ang.cos()[None, None], ang.sin()[None, None]) if pos.dim() == 1 else (ang.cos()[:, None], ang.sin()[:, None])
    x1, x2 = x.float()[..., : d // 2], x.float()[..., d // 2:]
    return torch.cat([x1 * cos - x2 * sin, x2 * cos + x1 * sin], dim=-1).to(x.dtype)


class Block(nn.Module):
    def __init__(self, s: LMShape, **kw):
        super().__init__()
        self.s = s
        kv = s.kv_heads * s.head_dim
        self.ln1, self.ln2 = RMSNorm(s.hidden, s.rms_eps, **kw), RMSNorm(s.hidden, s.rms_eps, **kw)
        self.q = nn.Linear(s.hidden, s.hidden, bias=True, **kw)
        self.k = nn.Linear(s.hidden, kv, bias=True, **kw)
        self.v = nn.Linear(s.hidden, kv, bias=True, **kw)
        self.o = nn.Linear(s.hidden, s.hidden, bias=False, **kw)
        self.gate = nn.Linear(s.hidden, s.intermediate, bias=False, **kw)
        self.up = nn.Linear(s.hidden, s.intermediate, bias=False, **kw)
        self.down = nn.Linear(s.intermediate, s.hidden, bias=False, **kw)

    def forward(self, x, pos, cache: Optional[Tuple[torch.Tensor, torch.Tensor]]):
        s = self.s
        B, T, _ = x.shape
        h = self.ln1(x)
        q = self.q(h).view(B, T, s.heads, s.head_dim).transpose(1, 2)
        k = self.k(h).view(B, T, s.kv_heads, s.head_dim).transpose(1, 2)
        v = self.v(h).view(B, T, s.kv_heads, s.head_dim).transpose(1, 2)
        q, k = _rope(q, pos, s.rope_theta), _rope(k, pos, s.rope_theta)
        if cache is not None:
            k = torch.cat([cache[0], k], dim=2)
            v = torch.cat([cache[1], v], dim=2)
        new_cache = (k, v)
        rep = s.heads // s.kv_heads
        kk = k.repeat_interleave(rep, dim=1) if rep > 1 else k
        vv = v.repeat_interleave(rep, dim=1) if rep > 1 else v
        S = kk.shape[2]
        # causal mask for T new queries sitting at the END of S keys
        mask = torch.ones(T, S, dtype=torch.bool, device=x.device).tril(diagonal=S - T)
        a = F.scaled_dot_product_attention(q, kk, vv, attn_mask=mask)
        x = x + self.o(a.transpose(1, 2).reshape(B, T, s.hidden))
        h = self.ln2(x)
        return x + self.down(F.silu(self.gate(h)) * self.up(h)), new_cache


    def forward_ragged(self, x, pos, kbuf, vbuf, window: int, rows=None):
        """Per-sequence positions (N3): x [B, T, D], pos [B, T] int64; kbuf / vbuf [B, Hkv, Tmax, hd] are
        written in place at `pos`, and query (b, t) attends keys j <= pos[b, t] of the first `window` slots.
        Entries past a sequence's committed length are never read before they are rewritten, so rolling a
        sequence back is a length update on the caller's side.  rows (int64 [B], optional): the cache rows
        these B sequences own, when only a subset of the cached sequences is fed (tier escalation)."""
        s = self.s
        B, T, _ = x.shape
        h = self.ln1(x)
        q = self.q(h).view(B, T, s.heads, s.head_dim).transpose(1, 2)
        k = self.k(h).view(B, T, s.kv_heads, s.head_dim).transpose(1, 2)
        v = self.v(h).view(B, T, s.kv_heads, s.head_dim).transpose(1, 2)
        q, k = _rope(q, pos, s.rope_theta), _rope(k, pos, s.rope_theta)
        bidx = (torch.arange(B, device=x.device) if rows is None else rows)[:, None].expand(B, T)
        kbuf[bidx, :, pos] = k.transpose(1, 2)
        vbuf[bidx, :, pos] = v.transpose(1, 2)
        if rows is None:
            kk, vv = kbuf[:, :, :window], vbuf[:, :, :window]
        else:
            kk, vv = kbuf[:, :, :window].index_select(0, rows), vbuf[:, :, :window].index_select(0, rows)
        rep = s.heads // s.kv_heads
        if rep > 1:
            kk, vv = kk.repeat_interleave(rep, dim=1), vv.repeat_interleave(rep, dim=1)
        mask = (torch.arange(window, device=x.device)[None, None, :] <= pos[:, :, None])[:, None]   # [B, 1, T, window]
        a = F.scaled_dot_product_attention(q, kk, vv, attn_mask=mask)
        x = x + self.o(a.transpose(1, 2).reshape(B, T, s.hidden))
        h = self.ln2(x)
        return x + self.down(F.silu(self.gate(h)) * self.up(h))


class SyntheticLM(nn.Module):
    """Decoder-only LM with a KV cache that can be rolled back after a rejection: `truncate` for the
    lock-step cache (one length for the batch), `alloc_ragged` / `forward_ragged` for per-sequence lengths."""

    def __init__(self, shape: LMShape, dtype: torch.dtype = torch.bfloat16, device=None, seed: int = 0,
                 logit_scale: float = 1.0):
        super().__init__()
        self.shape = shape
        self.logit_scale = logit_scale
        # parameters are created and initialised directly on the target device in the target dtype:
        # a 72B-shape model is 145 GB and must never pass through host memory
        dev = torch.device(device) if device is not None else torch.device("cpu")
        kw = dict(device="meta", dtype=dtype)          # meta + target dtype: to_empty() then allocates bf16 directly
        self.embed = nn.Embedding(shape.vocab, shape.hidden, **kw)
        self.blocks = nn.ModuleList([Block(shape, **kw) for _ in range(shape.layers)])
        self.norm = RMSNorm(shape.hidden, shape.rms_eps, **kw)
        self.lm_head = nn.Linear(shape.hidden, shape.vocab, bias=False, **kw)
        self.to_empty(device=dev)
        g = torch.Generator(device=dev).manual_seed(seed)
        with torch.no_grad():
            for name, p in self.named_parameters():
                if p.dim() >= 2:                       # matrices ~ N(0, 1/fan_in); norms 1, biases 0
                    p.normal_(0.0, 1.0 / math.sqrt(p.shape[-1]), generator=g)
                elif name.endswith(".bias"):
                    p.zero_()
                else:
                    p.fill_(1.0)
        self.eval()
        self._cache: List[Optional[Tuple[torch.Tensor, torch.Tensor]]] = [None] * shape.layers
        self._len = 0
        self._ragged = None
        self._graphs = None                    # enable_graphs(): {(B, T, return_hidden): (hipGraph, ids, pos0, out)}
        self._hip = None                       # enable_hip_layers(): serving.hip_decoder.HipDecoder

    @property
    def execution(self) -> str:
        """Which code runs forward_ragged: "hip_decoder" (asd_decoder_forward) or "torch_modules"."""
        return "hip_decoder" if self._hip is not None else "torch_modules"

    # -- cache management
    def reset(self):
        self._cache = [None] * self.shape.layers
        self._len = 0
        self._ragged = None
        if self._graphs is not None:
            self._graphs = {}

    def enable_graphs(self, on: bool = True, max_graphs: int = 4) -> None:
        """Replay full-batch forward_ragged calls (rows=None) from a hipGraph per (B, T) -- plumbing for the token-level
        loop: an eager pass of the 7B shape is ~1500 launches (15.9 ms at B = 32, T = 1, against 2.8 ms of weight
        streaming).  The graph attends over the WHOLE cache (positions past a sequence's length are masked as always), so
        its shapes do not depend on the step; inputs are copied into static buffers, the result is a static buffer that
        the next replay overwrites.  Subset feeds (rows given) and more than `max_graphs` distinct shapes stay eager."""
        self._graphs = {} if on else None
        self._max_graphs = int(max_graphs)

    def enable_hip_layers(self, on: bool = True, pack_weights: bool = False) -> None:
        """Run forward_ragged through asd_decoder_forward (X3: csrc/decoder.hip + asd_linear; nine launches per layer, one host
        call per pass) instead of the torch modules.  CUDA + bf16 + head_dim 128 only; raises otherwise -- no fallback.  Call
        before alloc_ragged.  The lock-step `forward` / `truncate` cache keeps the torch modules."""
        if getattr(self, "_weights_relaid", False) and not on:
            raise RuntimeError("this model's projection matrices were re-laid tile-major for the HIP decoder stack: the torch modules cannot run it any more")
        if on:
            from .hip_decoder import HipDecoder
            self._hip = HipDecoder(self, pack_weights=pack_weights)
            self._weights_relaid = self._hip.packed
        else:
            self._hip = None
        self._ragged = None

    def alloc_ragged(self, batch: int, max_len: int):
        """Per-sequence KV cache (N3): one [B, Hkv, max_len, hd] K and V buffer per layer, zero-filled.  Graphs captured by
        enable_graphs hold the OLD caches' addresses: they are dropped here (and re-captured on their next use)."""
        if self._graphs:
            self._graphs = {}
        if self._hip is not None:
            self._hip.alloc(batch, max_len)
            self._ragged = "hip"
            return
        p = next(self.parameters())
        shape = (batch, self.shape.kv_heads, max_len, self.shape.head_dim)
        self._ragged = [(torch.zeros(shape, dtype=p.dtype, device=p.device), torch.zeros(shape, dtype=p.dtype, device=p.device))
                        for _ in range(self.shape.layers)]

    @torch.no_grad()
    def forward_ragged(self, ids: torch.Tensor, pos0: torch.Tensor, window: int, return_hidden: bool = False,
                       rows: Optional[torch.Tensor] = None):
        """ids [B, T] placed at positions pos0[b] .. pos0[b]+T-1 of sequence b (pos0: [B] integer tensor);
        `window` is a host-side upper bound on any position in use (no device read-back).  KV entries at
        those positions are (re)written; nothing else changes.  Returns logits [B, T, V] (or hidden states).
        rows: cache rows of the B sequences when they are a subset of the allocated batch.  Positions past the
        cache (padding behind a ragged feed) are clamped into its last slot, which no real token ever uses."""
        assert self._ragged is not None, "call alloc_ragged first"
        if self._graphs is not None and rows is None and ids.is_cuda:
            key = (ids.shape[0], ids.shape[1], bool(return_hidden))
            if key in self._graphs or len(self._graphs) < self._max_graphs:
                return self._forward_ragged_graphed(key, ids, pos0, return_hidden)
        return self._forward_ragged_eager(ids, pos0, window, return_hidden, rows)

    def _cap(self) -> int:
        return self._hip.t_max if self._hip is not None else self._ragged[0][0].shape[2]

    def _forward_ragged_graphed(self, key, ids, pos0, return_hidden):
        g = self._graphs.get(key)
        if g is None:
            cap = self._cap()
            sid, spos = ids.clone(), pos0.to(torch.int64).clone()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):                 # warm-up outside the capture (library handles, autotuning)
                for _ in range(2):
                    self._forward_ragged_eager(sid, spos, cap, return_hidden, None)
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                out = self._forward_ragged_eager(sid, spos, cap, return_hidden, None)
            g = self._graphs[key] = (graph, sid, spos, out)
        graph, sid, spos, out = g
        sid.copy_(ids)
        spos.copy_(pos0)
        graph.replay()
        return out

    def _forward_ragged_eager(self, ids, pos0, window, return_hidden, rows):
        if self._hip is not None:
            return self._hip.forward(ids, pos0, return_hidden, rows)
        B, T = ids.shape
        cap = self._ragged[0][0].shape[2]
        window = min(window, cap)
        pos = (pos0.to(torch.int64)[:, None] + torch.arange(T, device=ids.device)).clamp_(max=cap - 1)
        x = self.embed(ids)
        for blk, (kb, vb) in zip(self.blocks, self._ragged):
            x = blk.forward_ragged(x, pos, kb, vb, window, rows)
        x = self.norm(x)
        if return_hidden:
            return x
        return self.lm_head(x) * self.logit_scale

    @property
    def cached_len(self) -> int:
        return self._len

    def truncate(self, length: int):
        """Keep the first `length` positions of the KV cache (KV rollback after a rejected suffix)."""
        if length >= self._len:
            return
        self._cache = [None if c is None else (c[0][:, :, :length], c[1][:, :, :length]) for c in self._cache]
        self._len = length

    @torch.no_grad()
    def forward(self, ids: torch.Tensor, return_hidden: bool = False) -> torch.Tensor:
        """ids: [B, T] NEW tokens (positions cached_len ... cached_len+T-1) -> logits [B, T, V].

        return_hidden=True stops before the lm_head and returns the final-norm states [B, T, D]: the input
        of asd_lm_head_verify (logits = hidden @ lm_head.weight.T * logit_scale are then never materialised)."""
        if getattr(self, "_weights_relaid", False):
            raise RuntimeError("projection matrices re-laid for the HIP decoder stack: use forward_ragged")
        B, T = ids.shape
        pos = torch.arange(self._len, self._len + T, device=ids.device)
        x = self.embed(ids)
        for i, blk in enumerate(self.blocks):
            x, self._cache[i] = blk(x, pos, self._cache[i])
        self._len += T
        x = self.norm(x)
        if return_hidden:
            return x
        return self.lm_head(x) * self.logit_scale
