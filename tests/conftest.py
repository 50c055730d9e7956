import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from tests import util
    return util.oracle()


@pytest.fixture(scope="session")
def hip_ctx():
    """A sah_ctx on cuda:0 sharing torch's current stream. Fails loudly if the HIP library is missing."""
    import torch
    from androidrenderer_amd import lib
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    ctx = lib.Context(device=0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    yield ctx
    torch.cuda.synchronize()
    ctx.close()
