import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def pytest_collection_finish(session):
    """Register the multi-rank GPU cases of the SELECTED tests so that tests/test_dist.py can run all cases of one rank count
    inside one torch.distributed.run launch (see the comment there)."""
    for item in session.items:
        spec = getattr(getattr(item, "function", None), "_dist_case", None)
        if spec is None or not hasattr(item, "callspec"):
            continue
        mod = sys.modules.get(item.function.__module__)
        case = spec(**item.callspec.params)
        if case is not None and mod is not None:
            mod.register_case(*case)
