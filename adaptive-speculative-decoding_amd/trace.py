"""roctx ranges around the hot-path calls of the host loop (the reference only has time.time() deltas,
pipeline.py:186,269; rocprofv3 --marker-trace / --kernel-trace then shows which tier step a kernel belongs to).

torch.cuda.nvtx IS roctx on PyTorch-ROCm.  Ranges cost ~100 ns each; ASD_ROCTX=0 switches them off.  If the marker library
is missing the ranges silently become no-ops -- they are diagnostics, never part of a result."""
from __future__ import annotations

import functools
import os

_state = {"on": None}


def enabled() -> bool:
    if _state["on"] is None:
        on = os.environ.get("ASD_ROCTX", "1") != "0"
        if on:
            try:
                import torch
                on = torch.cuda.is_available()
                if on:
                    torch.cuda.nvtx.range_push("asd:init")
                    torch.cuda.nvtx.range_pop()
            except Exception:  # noqa: BLE001
                on = False
        _state["on"] = bool(on)
    return _state["on"]


class range:  # noqa: A001  (context manager named after what it opens)
    __slots__ = ("name", "live")

    def __init__(self, name: str):
        self.name, self.live = name, False

    def __enter__(self):
        if enabled():
            import torch
            torch.cuda.nvtx.range_push(self.name)
            self.live = True
        return self

    def __exit__(self, *exc):
        if self.live:
            import torch
            torch.cuda.nvtx.range_pop()
        return False


def traced(name: str):
    """Decorator: the call runs inside the roctx range `asd:<name>`."""
    def deco(fn):
        @functools.wraps(fn)
        def wrapper(*a, **k):
            with range("asd:" + name):
                return fn(*a, **k)
        return wrapper
    return deco
