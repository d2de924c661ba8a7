import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950) device; run with -m gpu")


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (test infrastructure), built on demand with gcc."""
    from oracle import oracle as _orc
    _orc.build()
    return _orc


@pytest.fixture(scope="session")
def gpu_ctx():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    from multimotionfusion_amd.cudafuncs import Context
    ctx = Context(0)
    assert ctx.device_name().startswith("gfx950"), ctx.device_name()
    yield ctx
    ctx.close()
