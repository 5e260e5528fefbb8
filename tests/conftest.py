import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
for p in (str(ROOT), str(ROOT / "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (runs the HIP kernels through the C ABI)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    oracle_lib.lib()
    return oracle_lib


@pytest.fixture(scope="session")
def gpu():
    """rlshaders_amd Context on cuda:0.  GPU tests must run the HIP library: a missing library or a
    missing device is a failure, never a skip."""
    import torch
    import rlshaders_amd as R
    assert torch.cuda.is_available(), "gpu-marked test run without a GPU"
    R.load()
    ctx = R.Context(0)
    yield ctx
    ctx.close()
