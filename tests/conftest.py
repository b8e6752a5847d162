"""pytest configuration: `gpu` marker, repo root on sys.path, golden-fixture loader."""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """Bring the native artefacts up to date with the sources (no-op when they are): the HIP library
    (hipcc cross-compiles without a GPU) and the oracle's C restatement."""
    import importlib.util
    try:
        spec = importlib.util.spec_from_file_location(
            "asd_amd_build", os.path.join(ROOT, "adaptive-speculative-decoding_amd", "build.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        mod.build()
    except Exception as e:  # noqa: BLE001  (the ABI tests will then fail loudly with the reason)
        print(f"[conftest] libasd_hip.so build failed: {e}", file=sys.stderr)


def pytest_collection_modifyitems(config, items):
    """GPU tests are skipped (not failed) when no device is visible, e.g. a plain `pytest tests/`."""
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:  # pragma: no cover
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def load_npz(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def load_json(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def golden():
    class G:
        npz = staticmethod(load_npz)
        json = staticmethod(load_json)
    return G
