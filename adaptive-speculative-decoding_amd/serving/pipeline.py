"""Adaptive cascade pipeline -- API of the reference's src/serving/pipeline.py, with the decision
arithmetic of the stage loop (Bayes adjustment, DP stop rule) done by batched kernels.

    PipelineConfig                  pipeline.py:22-31   (+ stop_rule, stage_names: build extensions)
    RequestResult                   pipeline.py:34-45
    AdaptiveSpeculativePipeline     pipeline.py:48-423
        process_request / process_request_async / batch_process / update_lambda / get_stats /
        reset_stats / warmup / shutdown

What the reference's loop does per request (pipeline.py:165-286), and what this one keeps:
  for each stage i:  outputs, logprobs = stage.generate(prompts=[prompt_i], ..., return_logprobs=True)
                     p_i = predictor.predict(...)  (1.0 at the last stage)        :225-241
                     p_i = bayesian_adjustment(p_i, max(100, total_requests), a, b)   :234-238
                     k*  = optimal_stopping_rule(p[:i+1], C[:i+1], lambda)            :251-256
                     stop when k* == i, else prompt_{i+1} = prompt + " " + output     :259-266

`batch_process` really batches: all still-active requests of a stage go through ONE stage.generate call, ONE Bayes
launch and ONE DP launch (`batch_grouping="none"`, the default: one batch, as the reference's call shape implies).
Opt-in `batch_grouping="predicted_stage"` implements the reference's TODO (:331-338: "intelligent batching based on
predicted stages"): the predictor is asked for every request's acceptance probability at every stage from the
PROMPT alone, ONE DP launch over [n, L] turns that into a predicted stop stage, requests are grouped by it and the
groups run shallowest first (a group's requests leave the cascade together, so the stage calls of a group stay
full and easy requests are not held back by hard ones; but it means up to L stage-0 calls on smaller batches and
n * (L - 1) prompt-only predictor calls up front -- one per stage with a predictor that offers `predict_batch` -- so it
is not the default).

stop_rule:
  "prefix" -- the reference's rule verbatim.  Because the DP is run on the prefix p[:i+1], and a
              one-stage problem always stops, this ALWAYS stops at stage 0.
  "full"   -- (default) the DP sees all L stages: observed probabilities for stages <= i, prior
              `stage_priors` for the later ones (1.0 for the last).  This is the behaviour the
              reference's paper describes.
The reference's `_process_stages` also references an undefined `start_time` (NameError, F5); the
latency here is measured from the start of the request.
"""
from __future__ import annotations

import asyncio
import logging
import threading
import time
import uuid
from concurrent.futures import ThreadPoolExecutor
from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional, Sequence

import numpy as np

from ..backend import get_backend
from .cache import RequestCache

logger = logging.getLogger(__name__)

DEFAULT_STAGE_NAMES = ("8b", "13b", "34b", "70b")          # pipeline.py:175


@dataclass
class PipelineConfig:
    lambda_value: float = 1.0
    risk_adjustment: bool = True
    risk_alpha: float = 1.0
    risk_beta: float = 1.0
    enable_caching: bool = True
    max_concurrent_requests: int = 100
    batch_timeout_ms: float = 50.0
    # --- build extensions (defaults keep the reference's call sites working) ---
    stop_rule: str = "full"                                 # "full" | "prefix" (see module docstring)
    stage_names: Sequence[str] = DEFAULT_STAGE_NAMES
    stage_priors: Optional[Sequence[float]] = None          # prior p for not-yet-run stages ("full")
    batch_grouping: str = "none"                            # batch_process: "none" (one batch, the reference's shape) | "predicted_stage"

    @classmethod
    def from_yaml(cls, path: str) -> "PipelineConfig":
        """Read the `pipeline:` section of a serving YAML (reference schema: configs/serving.yaml:10-30;
        server.py:97-99 reads the same file raw).  Unknown keys are ignored."""
        import yaml
        with open(path, "r") as f:
            sec = (yaml.safe_load(f) or {}).get("pipeline", {}) or {}
        risk = sec.get("risk_adjustment", {})
        if isinstance(risk, bool):
            risk = {"enabled": risk}
        kw = dict(lambda_value=float(sec.get("lambda_value", 1.0)), risk_adjustment=bool(risk.get("enabled", True)),
                  risk_alpha=float(risk.get("alpha", 1.0)), risk_beta=float(risk.get("beta", 1.0)),
                  enable_caching=bool((sec.get("cache", {}) or {}).get("enable_kv_cache", True)),
                  batch_timeout_ms=float((sec.get("batching", {}) or {}).get("batch_timeout_ms", 50.0)))
        if "max_concurrent_requests" in sec:
            kw["max_concurrent_requests"] = int(sec["max_concurrent_requests"])
        if "stop_rule" in sec:
            kw["stop_rule"] = str(sec["stop_rule"])
        if "stage_names" in sec:
            kw["stage_names"] = tuple(sec["stage_names"])
        if "stage_priors" in sec:
            kw["stage_priors"] = tuple(float(x) for x in sec["stage_priors"])
        if "batch_grouping" in sec:
            kw["batch_grouping"] = str(sec["batch_grouping"])
        return cls(**kw)


@dataclass
class RequestResult:
    request_id: str
    output: str
    stopped_at_stage: int
    latency_ms: float
    stage_probabilities: List[float]
    stage_costs: List[float]
    cache_hits: int
    total_tokens: int
    tokens_per_second: float
    # --- build extensions (trailing, defaulted: the reference's positional construction still works) ---
    stages_run: int = 0                                     # stages actually executed (>= stopped_at_stage + 1)
    executed_costs: List[float] = field(default_factory=list)   # cost_per_token of every executed stage
    predicted_stage: int = -1                               # batch_process grouping key (-1: not predicted)


@dataclass
class _Active:
    """Book-keeping of one in-flight request inside a batch."""
    request_id: str
    prompt: str
    current_prompt: str
    start_time: float
    probabilities: List[float] = field(default_factory=list)
    costs: List[float] = field(default_factory=list)
    outputs: List[str] = field(default_factory=list)
    cache_hits: int = 0
    total_tokens: int = 0
    k_star: int = -1
    predicted_stage: int = -1


def _fresh_stats(n_stages: int) -> Dict[str, Any]:
    return {"total_requests": 0, "stage_stops": [0] * n_stages, "avg_latency": 0.0,
            "avg_tokens_per_second": 0.0, "total_tokens": 0, "avg_stage_probabilities": [0.0] * n_stages,
            "error_count": 0}


class AdaptiveSpeculativePipeline:
    def __init__(self, stage_manager, predictor, feature_extractor, config: PipelineConfig, cache_manager=None):
        self.stage_manager = stage_manager
        self.predictor = predictor
        self.feature_extractor = feature_extractor
        self.config = config
        if cache_manager is None and config.enable_caching:
            cache_manager = RequestCache()
        self.cache_manager = cache_manager
        self._n_stages = len(config.stage_names)
        self.stats = _fresh_stats(max(4, self._n_stages))
        self._stats_lock = threading.Lock()            # the reference mutates stats from 100 threads unlocked
        self.executor = ThreadPoolExecutor(max_workers=config.max_concurrent_requests)
        self.active_requests: Dict[str, Dict[str, Any]] = {}
        if config.stop_rule not in ("full", "prefix"):
            raise ValueError("stop_rule must be 'full' or 'prefix'")
        if config.batch_grouping not in ("predicted_stage", "none"):
            raise ValueError("batch_grouping must be 'predicted_stage' or 'none'")
        logger.info("AdaptiveSpeculativePipeline initialized")

    # ------------------------------------------------------------------ public API
    def process_request(self, prompt: str, max_tokens: int = 512, temperature: float = 0.7,
                        request_id: Optional[str] = None) -> RequestResult:
        return self._run_batch([prompt], max_tokens, temperature, [request_id])[0]

    async def process_request_async(self, prompt: str, max_tokens: int = 512, temperature: float = 0.7,
                                    request_id: Optional[str] = None) -> RequestResult:
        loop = asyncio.get_event_loop()
        return await loop.run_in_executor(self.executor, self.process_request, prompt, max_tokens, temperature,
                                          request_id)

    def batch_process(self, prompts: List[str], max_tokens: int = 512, temperature: float = 0.7) -> List[RequestResult]:
        prompts = list(prompts)
        if self.config.batch_grouping != "predicted_stage" or len(prompts) < 2 or self.config.stop_rule == "prefix":
            return self._run_batch(prompts, max_tokens, temperature, [None] * len(prompts))
        pred = self.predict_stop_stages(prompts)
        results: List[Optional[RequestResult]] = [None] * len(prompts)
        for stage in sorted(set(pred.tolist())):                       # shallowest group first
            idx = [i for i, s in enumerate(pred) if s == stage]
            out = self._run_batch([prompts[i] for i in idx], max_tokens, temperature, [None] * len(idx),
                                  predicted=[int(stage)] * len(idx))
            for i, r in zip(idx, out):
                results[i] = r
        return results  # type: ignore[return-value]

    def predict_stop_stages(self, prompts: Sequence[str]) -> np.ndarray:
        """The grouping key of batch_process: the stage the DP rule would stop at if every stage's acceptance
        probability were what the predictor says from the prompt alone (no output, no log-probs yet).  One Bayes
        launch and ONE DP launch for all n requests x L stages."""
        cfg = self.config
        names = list(cfg.stage_names)
        L = len(names)
        P = np.ones((len(prompts), L), dtype=np.float64)
        batched = getattr(self.predictor, "predict_batch", None)
        for i in range(L - 1):
            if batched is not None:                                     # ONE predictor forward per stage for the whole batch
                P[:, i] = np.asarray(batched(prompts=list(prompts), draft_outputs=[""] * len(prompts), draft_logprobs=None,
                                             stage_id=i, feature_extractor=self.feature_extractor), dtype=np.float64)
            else:                                                       # the reference's per-request predictor interface
                for j, prompt in enumerate(prompts):
                    P[j, i] = float(self.predictor.predict(prompt=prompt, draft_output="", draft_logprobs=None, stage_id=i,
                                                           feature_extractor=self.feature_extractor))
        backend = get_backend()
        if cfg.risk_adjustment:
            with self._stats_lock:
                n_obs = max(100, self.stats["total_requests"])
            P[:, :L - 1] = backend.bayes_adjust(P[:, :L - 1].reshape(-1), n_obs, cfg.risk_alpha, cfg.risk_beta).reshape(-1, L - 1)
        costs = np.array([float(self.stage_manager.get_stage(n).cost_per_token) for n in names])
        k_star, _ = backend.optimal_stopping(P, costs, cfg.lambda_value, False, 1.0, 1.0)
        return np.asarray(k_star, dtype=np.int64)

    def update_lambda(self, new_lambda: float):
        old = self.config.lambda_value
        self.config.lambda_value = new_lambda
        logger.info("Updated lambda: %.3f -> %.3f", old, new_lambda)

    def get_stats(self) -> Dict[str, Any]:
        with self._stats_lock:
            stats = dict(self.stats)
            stats["stage_stops"] = list(stats["stage_stops"])
            stats["avg_stage_probabilities"] = list(stats["avg_stage_probabilities"])
        n = stats["total_requests"]
        if n > 0:
            stats["stage_distribution"] = [c / n for c in stats["stage_stops"]]
            stats["avg_tokens_per_request"] = stats["total_tokens"] / n
        else:
            stats["stage_distribution"] = [0.0] * len(stats["stage_stops"])
            stats["avg_tokens_per_request"] = 0.0
        if self.cache_manager:
            stats["cache_stats"] = self.cache_manager.get_stats()
        stats["active_requests"] = len(self.active_requests)
        # telemetry of the token-level hot path behind the stages, when one is attached ("keep get_stats() keys;
        # add hbm_gbps, kernel_us"): the last verify step of a profiling HipOps
        hot = getattr(self, "hot_path_ops", None)
        if hot is not None:
            stats.update({k: v for k, v in hot.stats().items() if k in ("kernel_us", "hbm_gbps")})
        return stats

    def attach_hot_path(self, ops) -> None:
        """ops: the distributed.HipOps (profile=True) the stages verify with; get_stats() then carries `kernel_us` and
        `hbm_gbps` of its last verify step."""
        self.hot_path_ops = ops

    def reset_stats(self):
        with self._stats_lock:
            self.stats = _fresh_stats(max(4, self._n_stages))
        logger.info("Pipeline statistics reset")

    def warmup(self, num_requests: int = 5):
        prompts = ["Hello, how are you today?", "What is the capital of France?",
                   "Explain machine learning in simple terms.", "Write a short poem about nature.",
                   "What are the benefits of renewable energy?"]
        for i in range(num_requests):
            try:
                self.process_request(prompt=prompts[i % len(prompts)], max_tokens=50, temperature=0.7)
            except Exception as e:  # noqa: BLE001  (reference: log and continue, pipeline.py:407-408)
                logger.warning("Warmup request %d failed: %s", i + 1, e)

    def shutdown(self):
        self.executor.shutdown(wait=True)
        logger.info("Pipeline shutdown completed")

    # ------------------------------------------------------------------ the stage loop, batched
    def _run_batch(self, prompts: List[str], max_tokens: int, temperature: float,
                   request_ids: List[Optional[str]], predicted: Optional[List[int]] = None) -> List[RequestResult]:
        now = time.time()
        reqs = [_Active(request_id=rid or str(uuid.uuid4()), prompt=p, current_prompt=p, start_time=now)
                for p, rid in zip(prompts, request_ids)]
        if predicted is not None:
            for r, s in zip(reqs, predicted):
                r.predicted_stage = s
        for r in reqs:
            self.active_requests[r.request_id] = {"start_time": r.start_time,
                                                  "prompt": r.prompt[:100] + "..." if len(r.prompt) > 100 else r.prompt}
        try:
            self._process_stages(reqs, max_tokens, temperature)
            results = [self._finish(r) for r in reqs]
            for res in results:
                self._update_stats(res)
            return results
        except Exception as e:
            logger.error("Batch of %d request(s) failed: %s", len(reqs), e)
            with self._stats_lock:
                self.stats["error_count"] += len(reqs)
            raise
        finally:
            for r in reqs:
                self.active_requests.pop(r.request_id, None)
                if self.cache_manager:
                    self.cache_manager.cleanup_request(r.request_id)

    def _predict(self, r: _Active, stage_idx: int, output: str, logprobs) -> float:
        return float(self.predictor.predict(prompt=r.current_prompt, draft_output=output, draft_logprobs=logprobs,
                                            stage_id=stage_idx, feature_extractor=self.feature_extractor))

    def _process_stages(self, reqs: List[_Active], max_tokens: int, temperature: float) -> None:
        cfg = self.config
        names = list(cfg.stage_names)
        L = len(names)
        backend = get_backend()
        active = list(reqs)
        for i, name in enumerate(names):
            if not active:
                break
            stage = self.stage_manager.get_stage(name)
            last_stage = i == L - 1
            # -- generation: cached outputs first, one generate() call for the rest
            todo, cached = [], {}
            for r in active:
                hit = self.cache_manager.get_cache(r.request_id, i) if self.cache_manager else None
                if hit:
                    r.cache_hits += 1                                              # any cached entry counts (pipeline.py:192-194)
                if hit and hit.get("output"):
                    cached[r.request_id] = hit["output"]
                else:
                    todo.append(r)
            gen_out: Dict[str, Any] = {}
            if todo:
                texts, logprobs, stage_stats = stage.generate(prompts=[r.current_prompt for r in todo],
                                                              max_tokens=max_tokens, temperature=temperature,
                                                              return_logprobs=True)
                for j, r in enumerate(todo):
                    lp = logprobs[j] if logprobs is not None and len(logprobs) > j else np.array([])
                    gen_out[r.request_id] = (texts[j], lp)
                    if self.cache_manager:
                        self.cache_manager.allocate(r.request_id, i, {"output": texts[j], "logprobs": lp})
                logger.debug("Stage %d: %d generated, time=%.1fms", i, len(todo),
                             float(stage_stats.get("generation_time_ms", 0.0)) if stage_stats else 0.0)
            # -- predictor (host objects, per request as in the reference) ...
            probs = np.ones(len(active), dtype=np.float64)
            for j, r in enumerate(active):
                text, lp = gen_out.get(r.request_id, (cached.get(r.request_id), np.array([])))
                r.outputs.append(text)
                r.costs.append(float(stage.cost_per_token))
                r.total_tokens += len(text.split())
                if not last_stage:
                    probs[j] = self._predict(r, i, text, lp if len(lp) else None)
            # ... then ONE Bayes launch and ONE DP launch for the whole batch
            if not last_stage and cfg.risk_adjustment:
                with self._stats_lock:
                    n_obs = max(100, self.stats["total_requests"])                 # pipeline.py:235
                probs = backend.bayes_adjust(probs, n_obs, cfg.risk_alpha, cfg.risk_beta)
            for j, r in enumerate(active):
                r.probabilities.append(float(probs[j]))
            if cfg.stop_rule == "prefix":
                P = np.array([r.probabilities for r in active], dtype=np.float64)       # [n, i+1]
                costs = np.array(active[0].costs, dtype=np.float64)
            else:
                P = np.ones((len(active), L), dtype=np.float64)
                if cfg.stage_priors is not None:
                    P[:, :] = np.asarray(cfg.stage_priors, dtype=np.float64)[None, :L]
                P[:, L - 1] = 1.0
                P[:, :i + 1] = np.array([r.probabilities for r in active], dtype=np.float64)
                costs = np.array([float(self.stage_manager.get_stage(n).cost_per_token) for n in names])
            k_star, _ = backend.optimal_stopping(P, costs, cfg.lambda_value, False, 1.0, 1.0)
            still = []
            for j, r in enumerate(active):
                r.k_star = int(k_star[j])
                if last_stage:
                    continue
                # "prefix": the reference's test verbatim (k* == i, pipeline.py:259-261).  "full": k* <= i also
                # stops -- the rule says the best stopping point is already behind us; its output is returned.
                stop = (r.k_star == i) if cfg.stop_rule == "prefix" else (r.k_star <= i)
                if stop:
                    continue
                r.current_prompt = r.prompt + " " + r.outputs[-1]                 # pipeline.py:266
                still.append(r)
            active = still

    def _finish(self, r: _Active) -> RequestResult:
        k = r.k_star if 0 <= r.k_star < len(r.outputs) else len(r.outputs) - 1
        total_ms = (time.time() - r.start_time) * 1000
        tps = r.total_tokens / (total_ms / 1000) if total_ms > 0 else 0
        if self.cache_manager:
            self.cache_manager.truncate_at_stage(r.request_id, k)
        return RequestResult(request_id=r.request_id, output=r.outputs[k], stopped_at_stage=k, latency_ms=total_ms,
                             stage_probabilities=r.probabilities, stage_costs=r.costs[:k + 1],
                             cache_hits=r.cache_hits, total_tokens=r.total_tokens, tokens_per_second=tps,
                             stages_run=len(r.outputs), executed_costs=list(r.costs), predicted_stage=r.predicted_stage)

    def _update_stats(self, result: RequestResult):
        a = 0.01                                                                   # pipeline.py:295
        with self._stats_lock:
            st = self.stats
            st["total_requests"] += 1
            if result.stopped_at_stage < len(st["stage_stops"]):
                st["stage_stops"][result.stopped_at_stage] += 1
            st["total_tokens"] += result.total_tokens
            st["avg_latency"] = (1 - a) * st["avg_latency"] + a * result.latency_ms
            st["avg_tokens_per_second"] = (1 - a) * st["avg_tokens_per_second"] + a * result.tokens_per_second
            for i, prob in enumerate(result.stage_probabilities):
                if i < len(st["avg_stage_probabilities"]):
                    st["avg_stage_probabilities"][i] = (1 - a) * st["avg_stage_probabilities"][i] + a * prob
