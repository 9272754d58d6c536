import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def _miopen_for_tests():
    """GPU runs: point MIOpen at a scratch COPY of the committed find-db and keep its reference ("naive") solvers out of the per-shape
    search (geodiffuser_amd/miopen_cache.py) — on a fresh box the first convolution of every new shape otherwise costs seconds, minutes
    over the full-width tests.  Must happen before the first convolution; harmless without a GPU."""
    try:
        import torch
        if not torch.cuda.is_available() or os.environ.get("MIOPEN_USER_DB_PATH"):
            return
        from geodiffuser_amd import miopen_cache
        miopen_cache.configure()            # scratch copy of the committed seed, removed at exit
    except Exception:  # noqa: BLE001 - an optimisation of the test run only
        pass


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    _miopen_for_tests()


def pytest_collection_modifyitems(config, items):
    """Without a GPU, gpu-marked tests are skipped (plain `pytest tests` then behaves like `-m "not gpu"`)."""
    try:
        import torch
        have_gpu = torch.cuda.is_available()
    except Exception:  # noqa: BLE001
        have_gpu = False
    if have_gpu:
        return
    skip = pytest.mark.skip(reason="needs a GPU (torch.cuda.is_available() is False)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
