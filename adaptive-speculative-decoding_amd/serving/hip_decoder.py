"""X3: the decoder stack of a SyntheticLM through asd_decoder_forward (csrc/decoder.hip + asd_linear) instead of torch modules.

The reference runs its tiers through transformers / vLLM (third party: src/serving/real_model_pipeline.py:135); the bench's
token-level loop needs SOME model execution around the path, and with torch modules a pass of the 7B shape was ~1500 launches.
This class keeps the SyntheticLM's parameters (same random weights, same storage: q|k|v and gate|up are re-pointed at fused
buffers) and replaces the per-layer module calls with nine HIP launches per layer issued from one native call.

There is no CPU fallback: constructing it without a CUDA device or without libasd_hip.so raises.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from .. import _binding as B
from .. import kernels as K


class HipDecoder:
    def __init__(self, lm, pack_weights: bool = False):
        """pack_weights: re-lay every projection matrix tile-major IN ITS OWN STORAGE (asd_lm_head_pack_weights: a column block's
        64-deep reduction step becomes one contiguous 32 KiB run instead of 256 strided lines; same results bit for bit, 3-7 %
        less time per projection).  Needs every matrix's row count to be a multiple of 256 (all Qwen2.5 shapes); the torch
        modules of this model then hold re-laid bytes and must not be used any more (SyntheticLM refuses)."""
        p = next(lm.parameters())
        if not p.is_cuda or p.dtype != torch.bfloat16:
            raise RuntimeError("HipDecoder needs a bf16 SyntheticLM on a CUDA device (there is no CPU path)")
        s = lm.shape
        if s.head_dim != 128:
            raise RuntimeError(f"HipDecoder: head_dim {s.head_dim} unsupported (128 only)")
        self.lm = lm
        self.lib = B.load_library()
        self.device = p.device
        dev = self.device
        self.inv_freq = (1.0 / (s.rope_theta ** (torch.arange(0, s.head_dim, 2, device=dev, dtype=torch.float32) / s.head_dim))).contiguous()
        self._fused = []
        with torch.no_grad():
            for blk in lm.blocks:
                # one buffer per fused projection; the modules' parameters become views of it (no second copy of the weights)
                qkv_w = torch.cat([blk.q.weight.data, blk.k.weight.data, blk.v.weight.data], dim=0).contiguous()
                qkv_b = torch.cat([blk.q.bias.data, blk.k.bias.data, blk.v.bias.data], dim=0).contiguous()
                h, kv = s.hidden, s.kv_heads * s.head_dim
                blk.q.weight.data, blk.k.weight.data, blk.v.weight.data = qkv_w[:h], qkv_w[h:h + kv], qkv_w[h + kv:]
                blk.q.bias.data, blk.k.bias.data, blk.v.bias.data = qkv_b[:h], qkv_b[h:h + kv], qkv_b[h + kv:]
                gu_w = torch.cat([blk.gate.weight.data, blk.up.weight.data], dim=0).contiguous()
                blk.gate.weight.data, blk.up.weight.data = gu_w[:s.intermediate], gu_w[s.intermediate:]
                self._fused.append((qkv_w, qkv_b, gu_w))
        self.packed = False
        if pack_weights:
            mats = [m for i, blk in enumerate(lm.blocks) for m in (self._fused[i][0], blk.o.weight.data, self._fused[i][2], blk.down.weight.data)]
            if all(m.shape[0] % 256 == 0 and m.shape[1] % 64 == 0 and m.is_contiguous() for m in mats):
                tmp = torch.empty(max(m.numel() for m in mats), dtype=torch.bfloat16, device=dev)
                for m in mats:
                    N, D = m.shape
                    assert int(self.lib.asd_lm_head_packed_bytes(N, D)) == m.numel() * 2
                    rc = self.lib.asd_lm_head_pack_weights(m.data_ptr(), D, B.DTYPE_BF16, N, D, tmp.data_ptr(), m.numel() * 2, K._stream())
                    B.check("asd_lm_head_pack_weights", rc)
                    m.view(-1).copy_(tmp[: m.numel()])
                del tmp
                self.packed = True
            else:
                raise RuntimeError("HipDecoder(pack_weights=True): a projection's row count is not a multiple of 256")
        self.t_max = 0
        self.k_cache = []
        self.vt_cache = []
        self._layers = None
        self._shape = None
        self._scratch: Optional[torch.Tensor] = None
        self._lin_ws = K.LinearWorkspace(dev, on_grow=self._drop_graphs)

    def _drop_graphs(self) -> None:
        """A buffer whose ADDRESS captured hipGraphs hold (KV caches, the decoder scratch, the linear workspace) is about to be
        replaced: the graphs of the owning SyntheticLM (enable_graphs) would write into memory the allocator may have handed to
        another tensor.  They are dropped and re-captured on their next use."""
        g = getattr(self.lm, "_graphs", None)
        if g:
            g.clear()

    # -- cache
    def alloc(self, batch: int, max_len: int) -> None:
        s = self.lm.shape
        self._drop_graphs()                        # the caches below replace the ones captured graphs point at
        self.t_max = (int(max_len) + 31) // 32 * 32
        kshape = (batch, s.kv_heads, self.t_max, s.head_dim)
        vshape = (batch, s.kv_heads, s.head_dim, self.t_max)
        self.k_cache = [torch.zeros(kshape, dtype=torch.bfloat16, device=self.device) for _ in range(s.layers)]
        self.vt_cache = [torch.zeros(vshape, dtype=torch.bfloat16, device=self.device) for _ in range(s.layers)]
        arr = (B.Layer * s.layers)()
        for i, blk in enumerate(self.lm.blocks):
            qkv_w, qkv_b, gu_w = self._fused[i]
            arr[i] = B.Layer(blk.ln1.weight.data_ptr(), qkv_w.data_ptr(), qkv_b.data_ptr(), blk.o.weight.data_ptr(),
                             blk.ln2.weight.data_ptr(), gu_w.data_ptr(), blk.down.weight.data_ptr(),
                             self.k_cache[i].data_ptr(), self.vt_cache[i].data_ptr(), 1 if self.packed else 0)
        self._layers = arr
        self._shape = B.DecoderShape(s.hidden, s.heads, s.kv_heads, s.head_dim, s.intermediate, float(s.rms_eps),
                                     self.inv_freq.data_ptr(), self.t_max)

    def _scratch_for(self, M: int) -> torch.Tensor:
        need = int(self.lib.asd_decoder_scratch_bytes(C.byref(self._shape), M))
        if need == 0:
            raise RuntimeError(f"asd_decoder_scratch_bytes: decoder shape {self.lm.shape} is not supported")
        if self._scratch is None or self._scratch.numel() < need:
            if self._scratch is not None:
                self._drop_graphs()
            self._scratch = torch.empty(need + 256, dtype=torch.uint8, device=self.device)
        return self._scratch

    # -- one pass
    @torch.no_grad()
    def forward(self, ids: torch.Tensor, pos0: torch.Tensor, return_hidden: bool, rows: Optional[torch.Tensor]):
        """ids [B, T] at positions pos0[b] .. pos0[b] + T - 1; rows: cache rows of the B sequences.  Positions >= t_max - 1 are
        clamped into the cache's LAST slot, which is a trash slot for the padding behind a ragged feed: callers size the cache one
        slot beyond the longest real sequence (hierarchy._SeqState.kv_slots does; alloc rounds up to a multiple of 32 on top)."""
        assert self._layers is not None, "call alloc first"
        lm, s = self.lm, self.lm.shape
        Bn, T = ids.shape
        M = Bn * T
        x = lm.embed(ids).view(M, s.hidden)
        pos = (pos0.to(torch.int32)[:, None] + torch.arange(T, device=ids.device, dtype=torch.int32)).clamp_(max=self.t_max - 1).reshape(M).contiguous()
        rows32 = None if rows is None else rows.to(torch.int32).contiguous()
        scratch = self._scratch_for(M)
        base = (scratch.data_ptr() + 255) // 256 * 256
        hn = torch.empty_like(x)
        rc = self.lib.asd_decoder_forward(self._layers, s.layers, C.byref(self._shape), x.data_ptr(), x.stride(0), pos.data_ptr(),
                                          None if rows32 is None else rows32.data_ptr(), Bn, T, lm.norm.weight.data_ptr(),
                                          hn.data_ptr(), hn.stride(0), base, scratch.numel() - (base - scratch.data_ptr()),
                                          K._stream())
        B.check("asd_decoder_forward", rc)
        if return_hidden:
            return hn.view(Bn, T, s.hidden)
        logits = K.linear(hn, lm.lm_head.weight, workspace=self._lin_ws)
        if lm.logit_scale != 1.0:
            logits = logits * lm.logit_scale
        return logits.view(Bn, T, s.vocab)
