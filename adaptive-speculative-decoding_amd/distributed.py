"""Multi-GPU forms of the verify step: one process per GPU, torch.distributed (backend "nccl" = RCCL
over xGMI on the GPU box, "gloo" in the CPU tests).  The reference has no collective of its own
(TP lives inside vLLM); these are the three placements SURVEY.md §8e derives from
the north star, and the rule they share is that FULL LOGITS NEVER CROSS A LINK:

  batch-parallel replicas   every rank verifies its own slice of the batch; no data-path
                            collective at all (optional all-gather of n_acc, 4 B per sequence).
  tiers on different GPUs   the verify kernel runs where the target logits are produced; the draft
                            rank ships tok + lp_d ([B,K] i32 + f32 = 2 KB at B=32, K=8) and gets
                            accept + n_acc back (~400 B): `TierLink`, point-to-point send/recv.
  vocab-sharded target      each rank reduces its [B,K,V/R] slice to (m2, s, g) triples
                            (asd_lse_partial), ONE all-gather of [B,K,3] f32 per rank (3 KB at B=32),
                            then every rank combines them in rank order (asd_accept_from_partials):
                            `VocabShardedVerifier`.  Moving the logits instead would cost 77.9 MB /
                            153 GB/s = 0.5 ms per hop against a ~17 us kernel.

The compute is reached through a small `ops` object so that the CPU test-suite can drive the
exchange logic over gloo with the oracle (tests/oracle_backend.py: OracleOps); the default is the
HIP kernels and there is no other implementation in this package.
"""
from __future__ import annotations

from typing import Tuple

import torch
import torch.distributed as dist

from .trace import traced


def shard_bounds(total: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous, balanced [start, end) of `total` items for `rank` of `world`."""
    return total * rank // world, total * (rank + 1) // world


class HipOps:
    """The product: device tensors through libasd_hip.so.

    Re-entrant: the reference serves requests from a pool of up to 100 threads (src/serving/pipeline.py:83,155), so the
    scratch a call needs -- verify workspaces, sampler mailboxes, lm_head partial buffers -- is kept PER CALLING THREAD
    (a workspace must never be shared by calls that may overlap; each thread launches on its own current stream).  Only
    the packed lm_head images are shared between threads: they are read-only snapshots of a weight matrix."""

    def __init__(self, pack_lm_head: bool = True, profile: bool = False):
        """profile: bracket every verify step (verify_accept / verify_stop / lm_head_verify) with a HIP event pair on the
        calling thread's stream; `stats()` then reports `kernel_us` / `hbm_gbps` of the last one (the keys SURVEY §5 adds
        to the reference's get_stats(), pipeline.py:346-370).  Off by default: an event pair costs 5-15 us on this stack.
        pack_lm_head: lm_head matrices handed to lm_head_verify / lm_head_partial are re-laid out once, tile-major
        (asd_lm_head_pack_weights: 3-12 % faster streaming, bit-identical results).  COST: one more copy of every such
        matrix in HBM (+V*D*2 bytes: +2.5 GB for the 152064 x 8192 head of the 72B tier, +1.1 GB for the 7B one); pass
        False where that memory is needed for KV or batch."""
        import threading
        from . import kernels
        self.K = kernels
        self._tls = threading.local()
        self._packed = {}                    # weight key -> packed image (shared, read-only)
        self._lock = threading.Lock()
        self.pack_lm_head = bool(pack_lm_head)
        self.profile = bool(profile)

    def _timed_step(self, what, nbytes, fn):
        if not self.profile:
            return fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn()
        e1.record()
        self._tls.last = (e0, e1, int(nbytes), what)       # per calling thread, like the scratch: stats() reports THIS thread's step
        return out

    def stats(self):
        """{"kernel_us", "hbm_gbps", "step"} of the last verify step the CALLING THREAD profiled (waits for it to finish); {} if none."""
        last = getattr(self._tls, "last", None)
        if last is None:
            return {}
        e0, e1, nbytes, what = last
        e1.synchronize()
        us = 1e3 * e0.elapsed_time(e1)
        return {"kernel_us": us, "hbm_gbps": nbytes / (us * 1e-6) / 1e9 if us > 0 else 0.0, "step": what,
                "algorithmic_bytes": nbytes}

    @property
    def _ws(self):
        d = getattr(self._tls, "ws", None)
        if d is None:
            d = self._tls.ws = {}
        return d

    def check_status(self) -> None:
        """The sticky status words (include/asd_hip.h: asd_workspace_status) of every hand-off workspace the CALLING THREAD has
        used -- verify workspaces, the draft and residual samplers' mailboxes -- in ONE synchronising read.  A kernel whose bounded
        wait for a hand-off word ran out poisoned what depended on it (lp_t = NaN and reject; score = NaN, k* = L - 1; tok = -1) and
        raised the word: this raises kernels.LostHandoffError for it, after re-initialising every workspace that reported a loss
        (the late word was never handed back empty, so such a workspace is no longer all-zero between calls).  The loop drivers
        call it once per step, at a point where they synchronise anyway."""
        wss = [w for w in self._ws.values() if isinstance(w, self.K._StatusWorkspace)]
        if not wss:
            return
        by_dev = {}
        for w in wss:
            by_dev.setdefault(w.buf.device, []).append(w)
        dirty = []
        for group in by_dev.values():
            words = torch.stack([w.status_word for w in group])
            if int(words.max().item()) != 0:
                dirty += [w for w, v in zip(group, words.tolist()) if v != 0]
        if dirty:
            names = ", ".join(f"{type(w).__name__} (status 0x{int(w.status_word.item()):x})" for w in dirty)
            for w in dirty:
                w.reset()
            raise self.K.LostHandoffError(f"a hand-off word never arrived in: {names}; the affected results were poisoned and the "
                                          "workspaces have been re-initialised")

    def _workspace(self, B, K, V, dtype, device):
        key = (B, K, str(dtype), str(device))
        ws = self._ws.get(key)
        if ws is None or ws.V < V:
            ws = self._ws[key] = self.K.VerifyWorkspace(B, K, V, dtype, device)
        return ws

    @traced("verify_accept")
    def verify_accept(self, logits, tok, lp_d, u, inv_temperature: float = 1.0):
        B, K = tok.shape
        ws = self._workspace(B, K, logits.shape[-1], logits.dtype, logits.device)
        nbytes = B * K * logits.shape[-1] * logits.element_size() + 17 * B * K + 12 * B
        r = self._timed_step("asd_verify_accept", nbytes,
                             lambda: self.K.verify_accept(logits, tok, lp_d, u, ws, inv_temperature=inv_temperature))
        return r.lp_target, r.accept, r.n_acc, r.accept_bits

    @traced("verify_stop")
    def verify_stop(self, logits, tok, lp_d, u, inv_temperature, pred, feat, p_hist, stage_idx, costs, lam,
                    risk_adjustment=True, n_obs=100, alpha=1.0, beta=1.0, stats_col=5):
        """ONE launch per tier step (asd_verify_accept_fused_ex): verify + accept, then -- inside the same kernel, by the
        wave that completes each sequence -- statistics of the K target log-probs -> predictor -> Bayes -> DP rule
        (pipeline.py:225-261).  Returns ((lp_t, accept, n_acc, bits), (score, k_star, p_hist)); p_hist is updated in
        place.  Bit-identical to verify_accept followed by predictor_stop (tests/test_gpu_predictor.py)."""
        B, K = tok.shape
        ws = self._workspace(B, K, logits.shape[-1], logits.dtype, logits.device)
        packed, in_dim, hidden = pred
        nbytes = B * K * logits.shape[-1] * logits.element_size() + 17 * B * K + 12 * B
        v, r = self._timed_step("asd_verify_accept_fused_ex", nbytes, lambda: self.K.verify_accept_fused(
            logits, tok, lp_d, u, ws, feat, packed, in_dim, hidden, stage_idx=stage_idx, L=p_hist.shape[1], stats_col=stats_col,
            risk_adjustment=risk_adjustment, n_obs=n_obs, alpha=alpha, beta=beta, p_hist=p_hist, Cc=costs, lam=lam,
            inv_temperature=inv_temperature))
        return (v.lp_target, v.accept, v.n_acc, v.accept_bits), (r.score, r.k_star, p_hist)

    @traced("lse_partial")
    def lse_partial(self, logits_shard, tok, v_offset, inv_temperature: float = 1.0):
        B, K = tok.shape
        ws = self._workspace(B, K, logits_shard.shape[-1], logits_shard.dtype, logits_shard.device)
        return self.K.lse_partial(logits_shard, tok, v_offset, ws, inv_temperature=inv_temperature)

    @traced("accept_from_partials")
    def accept_from_partials(self, msg_all, lp_d, u, inv_temperature: float = 1.0):
        r = self.K.accept_from_partials(msg_all, lp_d, u, inv_temperature=inv_temperature)
        return r.lp_target, r.accept, r.n_acc, r.accept_bits

    @staticmethod
    def _weight_key(w):
        """Identity of a weight matrix AS DATA: address, geometry, dtype and torch's in-place version counter.  The address
        alone is not enough: an in-place update (`w.add_`, a checkpoint reload into the same storage) keeps it,
        and a row slice of a larger matrix shares it."""
        return (w.data_ptr(), tuple(w.shape), tuple(w.stride()), str(w.dtype), str(w.device), int(w._version))

    def _lm_head(self, weight, B, K):
        wkey = self._weight_key(weight)
        key = ("lmh", wkey[0], K)
        ver = self._ws.get(key)
        if ver is not None and (ver.wkey != wkey or ver.B < B):
            ver = None                           # the matrix changed in place (repack), or a larger batch than it was sized for
        if ver is None:
            image = None
            can_pack = self.pack_lm_head and weight.shape[1] % 64 == 0
            if can_pack:
                with self._lock:
                    for k2 in [k2 for k2 in self._packed if k2[:5] == wkey[:5] and k2 != wkey]:
                        del self._packed[k2]     # older versions of this very matrix (a row slice at the same address is its own entry)
                    image = self._packed.get(wkey)
            ver = self.K.LmHeadVerifier(weight, B, K, packed=can_pack and image is None, packed_image=image)
            ver.wkey = wkey
            if can_pack and image is None:
                with self._lock:
                    self._packed.setdefault(wkey, ver.packed)
            self._ws[key] = ver
        return ver

    @traced("lm_head_partial")
    def lm_head_partial(self, hidden, weight_shard, tok, v_offset, inv_temperature: float = 1.0):
        B, K = tok.shape
        return self._lm_head(weight_shard, B, K).partial(hidden, tok, v_offset, inv_temperature)

    @traced("lm_head_verify")
    def lm_head_verify(self, hidden, weight, tok, lp_d, u, inv_temperature: float = 1.0):
        """N2: verify from hidden states [n,K,D] and the [V,D] lm_head matrix (logits stay in MFMA accumulators)."""
        B, K = tok.shape
        ver = self._lm_head(weight, B, K)
        nbytes = weight.numel() * weight.element_size() + B * K * weight.shape[1] * weight.element_size()
        r = self._timed_step("asd_lm_head_verify", nbytes, lambda: ver(hidden, tok, lp_d, u, inv_temperature=inv_temperature))
        return r.lp_target, r.accept, r.n_acc, r.accept_bits

    # -- the rest of a tier step (serving/hierarchy.py): stop decision, proposal, commit draw, bookkeeping
    def pack_predictor(self, predictor, device):
        """predictor: MinimalQualityPredictor-like (weights_numpy / input_dim / hidden_dim) -> opaque handle."""
        return (self.K.pack_mlp_weights(*predictor.weights_numpy(), device=device), predictor.input_dim, predictor.hidden_dim)

    @traced("predictor_stop")
    def predictor_stop(self, pred, lp, feat, p_hist, stage_idx, costs, lam, risk_adjustment=True, n_obs=100, alpha=1.0,
                       beta=1.0, stats_col=5):
        """asd_predictor_stop on [n] sequences: p_hist [n,L] f64 is updated in place (column stage_idx) and returned
        with (score f32 [n], k_star i32 [n])."""
        packed, in_dim, hidden = pred
        r = self.K.predictor_stop(feat, packed, in_dim, hidden, stage_idx=stage_idx, L=p_hist.shape[1], lp=lp,
                                  stats_col=stats_col, risk_adjustment=risk_adjustment, n_obs=n_obs, alpha=alpha, beta=beta,
                                  p_hist=p_hist, Cc=costs, lam=lam)
        return r.score, r.k_star, p_hist

    def _sampler(self, kind, B, V, dtype, device):
        key = (kind, V, str(dtype), str(device))
        s = self._ws.get(key)
        if s is None or s.B < B:
            cls = self.K.DraftSampler if kind == "draft" else self.K.ResidualSampler
            s = self._ws[key] = cls(B, V, dtype, device)
        return s

    @traced("draft_sample")
    def draft_sample(self, logits, r, inv_temperature: float = 1.0, top_p: float = 1.0):
        """X1: logits [B,V] -> (tok i32 [B], log q(tok) f32 [B], nucleus threshold f32 [B])."""
        d = self._sampler("draft", logits.shape[0], logits.shape[1], logits.dtype, logits.device)(logits, r, inv_temperature, top_p)
        return d.tok, d.lp, d.thr

    @traced("residual_sample")
    def residual_sample(self, t_logits, d_logits, n_acc, r, bonus, inv_temperature: float = 1.0, d_threshold=None):
        """t_logits / d_logits [n,K,V], bonus [n,V], n_acc i32 [n], r f32 [n] -> committed token i32 [n]."""
        s = self._sampler("residual", t_logits.shape[0], t_logits.shape[2], t_logits.dtype, t_logits.device)
        return s(t_logits, d_logits, n_acc, r, bonus, inv_temperature, d_threshold=d_threshold)

    @traced("commit_step")
    def commit_step(self, tok, n_acc, drawn, seq_len, tokens, n_commit, max_len):
        self.K.commit_step(tok, n_acc, drawn, seq_len, tokens, n_commit, max_len=max_len)

    @traced("lambda_sweep")
    def lambda_sweep(self, p_hist, costs, lams):
        """N4: the DP rule for every (lambda, sequence) pair in one launch -> k_star [G, n] i32."""
        return self.K.lambda_sweep(p_hist, costs, lams)[0]


def _world(group) -> Tuple[int, int]:
    return dist.get_world_size(group), dist.get_rank(group)


def host_staged(group=None) -> bool:
    """True when the group's backend cannot move device tensors itself (gloo: the CPU tests, and the one-GPU
    rehearsal of the multi-rank paths); messages are then staged through host memory.  RCCL moves them directly."""
    return dist.get_backend(group) == "gloo"


def all_gather_any(parts_like: torch.Tensor, group=None):
    """all_gather of equally shaped tensors -> list in rank order; device tensors over gloo go through the host."""
    world = dist.get_world_size(group)
    if parts_like.is_cuda and host_staged(group):
        src = parts_like.cpu()
        parts = [torch.empty_like(src) for _ in range(world)]
        dist.all_gather(parts, src, group=group)
        return [p.to(parts_like.device) for p in parts]
    parts = [torch.empty_like(parts_like) for _ in range(world)]
    dist.all_gather(parts, parts_like.contiguous(), group=group)
    return parts


def all_to_all_rows(rows: torch.Tensor, counts, group=None) -> torch.Tensor:
    """rows [sum(counts), W]: rows [o_q, o_q + counts[q]) go to rank q; returns [world * counts[me], W] -- the pieces every rank
    held of MY rows, in rank order.  One all-to-all (RCCL: ncclSend / ncclRecv pairs in one group); over gloo through the host."""
    world, me = dist.get_world_size(group), dist.get_rank(group)
    counts = [int(c) for c in counts]
    staged = rows.is_cuda and host_staged(group)
    src = rows.contiguous().cpu() if staged else rows.contiguous()
    out = torch.empty((world * counts[me],) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    dist.all_to_all_single(out, src, output_split_sizes=[counts[me]] * world, input_split_sizes=counts, group=group)
    return out.to(rows.device) if staged else out


class VocabShardedVerifier:
    """Target lm_head split over the ranks of `group` along the vocabulary."""

    def __init__(self, vocab: int, ops=None, group=None, inv_temperature: float = 1.0):
        self.vocab = vocab
        self.inv_temperature = float(inv_temperature)
        self.ops = ops if ops is not None else HipOps()
        self.group = group
        self.world, self.rank = _world(group)
        self.v0, self.v1 = shard_bounds(vocab, self.world, self.rank)

    def verify(self, logits_shard: torch.Tensor, tok: torch.Tensor, lp_d: torch.Tensor, u: torch.Tensor):
        """logits_shard: [B,K,v1-v0] of THIS rank; tok: GLOBAL ids.  Returns (lp_t, accept, n_acc, bits),
        identical on every rank (fixed combine order)."""
        if logits_shard.shape[-1] != self.v1 - self.v0:
            raise ValueError(f"rank {self.rank} expects a shard of width {self.v1 - self.v0}")
        msg = self.ops.lse_partial(logits_shard, tok, self.v0, self.inv_temperature).contiguous()
        return self._finish(msg, lp_d, u)

    def verify_hidden(self, hidden: torch.Tensor, weight_shard: torch.Tensor, tok: torch.Tensor, lp_d: torch.Tensor,
                      u: torch.Tensor):
        """Tensor-parallel lm_head (N2): `weight_shard` [v1-v0, D] is THIS rank's rows of the lm_head matrix,
        `hidden` [B,K,D] is replicated.  Each rank reduces its shard to the (m2, s, g) message without forming
        logits (asd_lm_head_partial); the exchange and the result are those of `verify`."""
        if weight_shard.shape[0] != self.v1 - self.v0:
            raise ValueError(f"rank {self.rank} expects a weight shard of {self.v1 - self.v0} rows")
        msg = self.ops.lm_head_partial(hidden, weight_shard, tok, self.v0, self.inv_temperature).contiguous()
        return self._finish(msg, lp_d, u)

    def _finish(self, msg, lp_d, u):
        parts = all_gather_any(msg, self.group)                # the one exchange step: [B,K,3] per rank
        return self.ops.accept_from_partials(torch.stack(parts).contiguous(), lp_d, u, self.inv_temperature)


class BatchShardedVerifier:
    """Batch-parallel replicas: rank r owns sequences [b0, b1); no data-path collective."""

    def __init__(self, batch: int, ops=None, group=None):
        self.ops = ops if ops is not None else HipOps()
        self.group = group
        self.world, self.rank = _world(group)
        self.batch = batch
        self.b0, self.b1 = shard_bounds(batch, self.world, self.rank)

    def verify_local(self, logits, tok, lp_d, u, inv_temperature: float = 1.0):
        """Arguments are THIS rank's rows [b0:b1)."""
        return self.ops.verify_accept(logits, tok, lp_d, u, inv_temperature)

    def gather_n_acc(self, n_acc_local: torch.Tensor) -> torch.Tensor:
        """Optional [B] view of the accepted lengths on every rank (token accounting)."""
        sizes = [shard_bounds(self.batch, self.world, r) for r in range(self.world)]
        width = max(e - s for s, e in sizes)
        pad = torch.zeros(width, dtype=n_acc_local.dtype, device=n_acc_local.device)
        pad[: n_acc_local.numel()] = n_acc_local
        parts = [torch.empty_like(pad) for _ in range(self.world)]
        dist.all_gather(parts, pad, group=self.group)
        return torch.cat([p[: e - s] for p, (s, e) in zip(parts, sizes)])


class TierLink:
    """Point-to-point hand-off between the rank that drafts and the rank that holds the target."""

    def __init__(self, draft_rank: int, target_rank: int, group=None):
        self.draft_rank, self.target_rank, self.group = draft_rank, target_rank, group

    # draft side --------------------------------------------------------------------------
    def send_draft(self, tok: torch.Tensor, lp_d: torch.Tensor) -> None:
        """tok [B,K] i32 + lp_d [B,K] f32 as ONE message (bit-cast into an i32 [2,B,K] buffer)."""
        buf = torch.stack([tok.to(torch.int32), lp_d.to(torch.float32).view(torch.int32)]).contiguous()
        dist.send(buf, dst=self.target_rank, group=self.group)

    def recv_verdict(self, B: int, K: int, device) -> Tuple[torch.Tensor, torch.Tensor]:
        buf = torch.empty(B * K + B, dtype=torch.int32, device=device)
        dist.recv(buf, src=self.target_rank, group=self.group)
        return buf[: B * K].view(B, K).to(torch.uint8), buf[B * K:].clone()

    # target side -------------------------------------------------------------------------
    def recv_draft(self, B: int, K: int, device) -> Tuple[torch.Tensor, torch.Tensor]:
        buf = torch.empty((2, B, K), dtype=torch.int32, device=device)
        dist.recv(buf, src=self.draft_rank, group=self.group)
        return buf[0].contiguous(), buf[1].contiguous().view(torch.float32)

    def send_verdict(self, accept: torch.Tensor, n_acc: torch.Tensor) -> None:
        buf = torch.cat([accept.reshape(-1).to(torch.int32), n_acc.reshape(-1).to(torch.int32)]).contiguous()
        dist.send(buf, dst=self.draft_rank, group=self.group)

    # generic small messages (committed tokens; ONE draft-logits row per sequence after a rejection)
    def send_to_target(self, t: torch.Tensor) -> None:
        dist.send(t.contiguous(), dst=self.target_rank, group=self.group)

    def send_to_draft(self, t: torch.Tensor) -> None:
        dist.send(t.contiguous(), dst=self.draft_rank, group=self.group)

    def recv_from_draft(self, shape, dtype, device) -> torch.Tensor:
        buf = torch.empty(shape, dtype=dtype, device=device)
        dist.recv(buf, src=self.draft_rank, group=self.group)
        return buf

    def recv_from_target(self, shape, dtype, device) -> torch.Tensor:
        buf = torch.empty(shape, dtype=dtype, device=device)
        dist.recv(buf, src=self.target_rank, group=self.group)
        return buf
