"""Per-request stage-output cache with the five methods the pipeline calls on its cache manager
(reference: src/serving/cache_manager.py:69,121,149,192,369 -- `allocate`, `get_cache`,
`truncate_at_stage`, `cleanup_request`, `get_stats`).

The reference's KVCacheManager is host bookkeeping of {"output": str, "logprobs": ndarray} entries
(not a real KV cache), with an LRU size limit and a janitor thread; it is outside the hot path
(out of scope, "next" row N3).  This class keeps the call surface so the pipeline
drops in, as a lock-guarded dict-of-dicts without the thread.
"""
from __future__ import annotations

import threading
import time
from typing import Any, Dict, Optional


class RequestCache:
    def __init__(self, max_entries: int = 65536):
        self._lock = threading.RLock()
        self._data: Dict[str, Dict[int, Dict[str, Any]]] = {}
        self._max_entries = max_entries
        self._stats = {"total_allocations": 0, "cache_hits": 0, "cache_misses": 0, "evictions": 0}

    def allocate(self, request_id: str, stage_id: int, cache_data: Dict[str, Any]) -> bool:
        with self._lock:
            if sum(len(v) for v in self._data.values()) >= self._max_entries:
                oldest = min(self._data, key=lambda r: min(e["_t"] for e in self._data[r].values()))
                self._stats["evictions"] += len(self._data.pop(oldest))
            self._data.setdefault(request_id, {})[stage_id] = {"_t": time.time(), **cache_data}
            self._stats["total_allocations"] += 1
            return True

    def get_cache(self, request_id: str, stage_id: int) -> Optional[Dict[str, Any]]:
        with self._lock:
            entry = self._data.get(request_id, {}).get(stage_id)
            self._stats["cache_hits" if entry is not None else "cache_misses"] += 1
            if entry is None:
                return None
            entry["_t"] = time.time()
            return {k: v for k, v in entry.items() if k != "_t"}

    def truncate_at_stage(self, request_id: str, stage_id: int) -> None:
        """Drop the entries of stages after `stage_id`."""
        with self._lock:
            stages = self._data.get(request_id)
            if stages:
                for s in [s for s in stages if s > stage_id]:
                    del stages[s]

    def cleanup_request(self, request_id: str) -> None:
        with self._lock:
            self._data.pop(request_id, None)

    def get_stats(self) -> Dict[str, Any]:
        with self._lock:
            out = dict(self._stats)
            out["active_requests"] = len(self._data)
            out["entries"] = sum(len(v) for v in self._data.values())
            return out
