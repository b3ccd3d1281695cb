import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def ctx():
    """Device context for -m gpu tests: fails (never skips to a CPU path) when the HIP library or device is missing."""
    import dfgpu
    return dfgpu.Context(0)


@pytest.fixture(scope="session")
def task_ctx(ctx):
    from dfgpu import physical_plan as ops
    return ops.TaskContext(ctx, batch_size=8192)
