import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with `pytest -m gpu`)")


@pytest.fixture(scope="session")
def hip_ctx():
    """A HIP context; fails loudly (never skips to a fallback) when the extension or the GPU is missing."""
    from uzu_amd import backend
    ctx = backend.Context.new(0)
    yield ctx
    ctx.close()
