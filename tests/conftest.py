import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def goldens():
    with open(os.path.join(ROOT, "tests", "golden", "reference_goldens.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def gpu_ctx_factory():
    """Creates device contexts; fails loudly (never skips) when the HIP library or GPU is missing."""
    from perphil_amd import _ffi

    made = []

    def make(device=0):
        ctx = _ffi.Context(device)
        made.append(ctx)
        return ctx

    yield make
    for c in made:
        c.close()
